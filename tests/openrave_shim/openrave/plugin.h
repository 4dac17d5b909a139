// Test double for <openrave/plugin.h> (tests/openrave_shim/README.md): the names plugin/mcsimplugin_pocs.cpp
// uses, with OpenRAVE 0.9's signatures and the least behaviour the tests need.  NOT OpenRAVE.
#pragma once
#include <cstdio>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define OPENRAVE_PLUGIN_API extern "C"
#define RAVELOG_ERROR(...) std::fprintf(stderr, "[shim error] " __VA_ARGS__)
#define RAVELOG_WARN(...) std::fprintf(stderr, "[shim warn] " __VA_ARGS__)
#define RAVELOG_INFO(...) std::fprintf(stderr, "[shim info] " __VA_ARGS__)

namespace OpenRAVE {

typedef double dReal;
struct Vector { dReal x, y, z, w; Vector(dReal X = 0, dReal Y = 0, dReal Z = 0) : x(X), y(Y), z(Z), w(0) {} };

// a rigid transform; OpenRAVE keeps a quaternion, the double keeps the rotation matrix (row-major 3 x 3)
struct Transform {
  dReal R[9];
  Vector trans;
  Transform() : R{1, 0, 0, 0, 1, 0, 0, 0, 1} {}
  Transform operator*(const Transform& o) const {
    Transform r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r.R[3 * i + j] = R[3 * i] * o.R[j] + R[3 * i + 1] * o.R[3 + j] + R[3 * i + 2] * o.R[6 + j];
    r.trans = Vector(R[0] * o.trans.x + R[1] * o.trans.y + R[2] * o.trans.z + trans.x,
                     R[3] * o.trans.x + R[4] * o.trans.y + R[5] * o.trans.z + trans.y,
                     R[6] * o.trans.x + R[7] * o.trans.y + R[8] * o.trans.z + trans.z);
    return r;
  }
};
struct TransformMatrix {          // OpenRAVE: m[12], three rows of the rotation with a stride of 4
  dReal m[12];
  Vector trans;
  explicit TransformMatrix(const Transform& t) : trans(t.trans) {
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m[4 * i + j] = t.R[3 * i + j]; m[4 * i + 3] = 0; }
  }
};
struct AABB { Vector pos, extents; };
enum GeometryType { GT_None = 0, GT_Box = 1, GT_Sphere = 2, GT_Cylinder = 3, GT_TriMesh = 4 };

class KinBody {
 public:
  class Link {
   public:
    class Geometry {
     public:
      GeometryType type = GT_Box;
      Transform t;
      Vector extents;
      GeometryType GetType() const { return type; }
      const Transform& GetTransform() const { return t; }
      const Vector& GetBoxExtents() const { return extents; }
    };
    typedef std::shared_ptr<Geometry> GeometryPtr;
    std::string name;
    Transform t;
    std::vector<GeometryPtr> geoms;
    AABB local;
    Transform GetTransform() const { return t; }
    const std::vector<GeometryPtr>& GetGeometries() const { return geoms; }
    const std::string& GetName() const { return name; }
    AABB ComputeLocalAABB() const { return local; }
  };
  typedef std::shared_ptr<Link> LinkPtr;
  virtual ~KinBody() {}
  std::string name;
  std::vector<LinkPtr> links;
  const std::vector<LinkPtr>& GetLinks() const { return links; }
  const std::string& GetName() const { return name; }
};
typedef std::shared_ptr<KinBody> KinBodyPtr;
class RobotBase : public KinBody {};
typedef std::shared_ptr<RobotBase> RobotBasePtr;

struct EnvironmentMutex : std::recursive_mutex { typedef std::unique_lock<std::recursive_mutex> scoped_lock; };
class EnvironmentBase {
 public:
  EnvironmentMutex mutex;
  std::vector<KinBodyPtr> bodies;
  std::vector<RobotBasePtr> robots;
  EnvironmentMutex& GetMutex() { return mutex; }
  void GetRobots(std::vector<RobotBasePtr>& out) const { out = robots; }
  void GetBodies(std::vector<KinBodyPtr>& out) const { out = bodies; }
};
typedef std::shared_ptr<EnvironmentBase> EnvironmentBasePtr;

struct openrave_exception : std::runtime_error { explicit openrave_exception(const std::string& s) : std::runtime_error(s) {} };

enum InterfaceType { PT_Planner = 1, PT_Robot = 2, PT_Module = 9 };
class InterfaceBase {
 public:
  typedef std::function<bool(std::ostream&, std::istream&)> InterfaceCommandFn;
  virtual ~InterfaceBase() {}
  void RegisterCommand(const std::string& name, InterfaceCommandFn fn, const std::string& help) { cmds_[name] = fn; help_[name] = help; }
  // InterfaceBase::SendCommand: the first token names the command, the handler gets the stream behind it
  virtual bool SendCommand(std::ostream& sout, std::istream& sinput) {
    std::string name;
    if (!(sinput >> name)) return false;
    if (name == "help") { for (auto& kv : help_) sout << kv.first << " - " << kv.second << "\n"; return true; }
    auto it = cmds_.find(name);
    if (it == cmds_.end()) return false;
    return it->second(sout, sinput);
  }
 private:
  std::map<std::string, InterfaceCommandFn> cmds_;
  std::map<std::string, std::string> help_;
};
typedef std::shared_ptr<InterfaceBase> InterfaceBasePtr;
class ModuleBase : public InterfaceBase {
 public:
  explicit ModuleBase(EnvironmentBasePtr penv) : env_(penv) {}
  EnvironmentBasePtr GetEnv() const { return env_; }
 private:
  EnvironmentBasePtr env_;
};
struct PLUGININFO { std::map<InterfaceType, std::vector<std::string> > interfacenames; };

}  // namespace OpenRAVE
