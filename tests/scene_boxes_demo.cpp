// scene_boxes_demo.cpp -- exercises csrc/scene_boxes.hpp (the geometry -> obstacle-table logic of the
// OpenRAVE adapter, plugin/mcsimplugin_pocs.cpp) without OpenRAVE and without a GPU.
//   usage: scene_boxes_demo geoms.txt        one geometry per line: name R00..R22 tx ty tz ex ey ez
//   prints "box cx cy hx hy yaw" per kept obstacle (the format of planio.load_env), "skipped ..." otherwise
#include <cstdio>
#include <fstream>
#include <sstream>

#include "../probability-of-collision-for-safe-planning_amd/csrc/scene_boxes.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  std::vector<pocs::BoxGeom> geoms;
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    pocs::BoxGeom g;
    is >> g.name;
    for (double& v : g.R) is >> v;
    for (double& v : g.t) is >> v;
    for (double& v : g.ext) is >> v;
    if (!is) return 3;
    geoms.push_back(g);
  }
  const pocs::SceneTable T = pocs::scene_to_table(geoms);
  for (int m = 0; m < T.M(); ++m)
    std::printf("box %.17g %.17g %.17g %.17g %.17g\n", T.boxes[5 * m], T.boxes[5 * m + 1], T.boxes[5 * m + 2],
                T.boxes[5 * m + 3], T.boxes[5 * m + 4]);
  for (const std::string& s : T.skipped) std::printf("skipped %s\n", s.c_str());
  return 0;
}
