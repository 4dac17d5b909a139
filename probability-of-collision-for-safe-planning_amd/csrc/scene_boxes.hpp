// scene_boxes.hpp -- from the box geometries of a loaded scene to the collision world of libpocs.
//
// The reference module receives its collision world through its constructor: `sim(penv)`
// (mcsimplugin/mcsimplugin.cpp:12) keeps the OpenRAVE environment and the first robot
// (MCSimulator.h:139-156) and asks `env->CheckCollision(robot)` per pose (:257-285).  The drop-in
// hands the same scene to libpocs once, as a table of planar oriented boxes
// (pocs_set_obstacles) and a footprint (pocs_set_footprint).  This header holds that conversion
// with no OpenRAVE type in it, so that it is compiled and tested here (tests/mcmodule_demo.cpp
// feeds it the geometries of the reference's scenes) and used verbatim by the OpenRAVE adapter
// (plugin/mcsimplugin_pocs.cpp), which only fills BoxGeom from KinBody::Link::Geometry.
// Same rules as the Python loader (envxml.py): boxes only; rotations about z only; a box is kept
// when its z range overlaps the robot's [z_min, z_max] (drops the floor and the door lintel).
#pragma once
#include <cmath>
#include <string>
#include <vector>

namespace pocs {

struct BoxGeom {
  double R[9];       // world rotation of the geometry, row-major (link transform x geometry transform)
  double t[3];       // world position of its centre
  double ext[3];     // half extents along its own axes
  std::string name;  // body/link, for messages
};

struct Obstacle { double cx, cy, hx, hy, yaw; };       // one row of pocs_set_obstacles

struct SceneTable {
  std::vector<double> boxes;          // M x 5, ready for pocs_set_obstacles
  std::vector<std::string> skipped;   // what was left out, and why
  int M() const { return (int)(boxes.size() / 5); }
};

// 0 = kept, 1 = outside the z range, 2 = not a rotation about z (cannot be represented on the plane)
inline int box_to_obstacle(const BoxGeom& g, double z_min, double z_max, Obstacle* out) {
  const double tol = 1e-9;
  // a rotation about z: third row and third column are (0, 0, 1)
  if (std::fabs(g.R[2]) > tol || std::fabs(g.R[5]) > tol || std::fabs(g.R[6]) > tol || std::fabs(g.R[7]) > tol ||
      std::fabs(g.R[8] - 1.0) > tol)
    return 2;
  if (g.t[2] + g.ext[2] < z_min || g.t[2] - g.ext[2] > z_max) return 1;
  out->cx = g.t[0]; out->cy = g.t[1];
  out->hx = g.ext[0]; out->hy = g.ext[1];
  out->yaw = std::atan2(g.R[3], g.R[0]);                 // R = Rz(yaw): R[0] = cos, R[3] = sin
  return 0;
}

inline SceneTable scene_to_table(const std::vector<BoxGeom>& geoms, double z_min = 0.05, double z_max = 1.5) {
  SceneTable T;
  for (const BoxGeom& g : geoms) {
    Obstacle o;
    const int why = box_to_obstacle(g, z_min, z_max, &o);
    if (why == 0) {
      const double row[5] = {o.cx, o.cy, o.hx, o.hy, o.yaw};
      T.boxes.insert(T.boxes.end(), row, row + 5);
    } else if (why == 2) {
      T.skipped.push_back(g.name + ": rotated out of the plane");
    }
  }
  return T;
}

// Footprint from the robot's base-link bounding box in the base frame (centre, half extents):
// pocs_set_footprint(dx, dy, hx, hy).
struct Footprint { double dx, dy, hx, hy; };
inline Footprint footprint_from_aabb(const double centre[3], const double half[3]) {
  Footprint f = {centre[0], centre[1], half[0], half[1]};
  return f;
}

}  // namespace pocs
