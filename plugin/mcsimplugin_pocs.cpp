// mcsimplugin_pocs.cpp -- the OpenRAVE plugin translation unit (drop-in for
// mcsimplugin/mcsimplugin.cpp): same interface name, same commands, estimator = libpocs.so.
// NOT built in this repository's image (OpenRAVE and Boost are absent); build it where they are:
//   g++ -shared -fPIC mcsimplugin_pocs.cpp $(openrave-config --cflags --libs-core) -I../include -lpocs
#include <openrave/plugin.h>
#include <boost/bind.hpp>
#include <cstdlib>

#include "../probability-of-collision-for-safe-planning_amd/csrc/mcmodule.hpp"

using namespace OpenRAVE;

class MCModule : public ModuleBase {
 public:
  MCModule(EnvironmentBasePtr penv, std::istream&) : ModuleBase(penv), impl_(0) {
    static const char* const names[] = {"MyCommand", "ArmaCommand", "setAlphas", "setQ", "setNumLandmarks",
        "setLandmarks", "setNumParticles", "setInitialCovariance", "setPathLength", "setTrajectory",
        "setOdometry", "runSimulation", "setNumGaussians", "runGMMEstimation", "setNumGMMSamples",
        "setSeed", "setFootprint", "addObstacle", "clearObstacles", "setBatch", "setRunAhead"};
    for (const char* n : names)
      RegisterCommand(n, boost::bind(&MCModule::Forward, this, std::string(n), _1, _2), "see include/pocs.h");
    // The reference driver issues one run* command per run, 200 in a row (MCSimulation.py:238-256):
    // evaluate them 16 at a time behind that interface (POCS_RUN_AHEAD overrides; 1 = off).
    const char* ra = getenv("POCS_RUN_AHEAD");
    impl_.SendCommand(std::string("setRunAhead ") + (ra ? ra : "16"));
  }
  bool Forward(const std::string& name, std::ostream& sout, std::istream& sinput) {
    std::stringstream line;
    line << name << ' ' << sinput.rdbuf();
    if (!impl_.SendCommand(sout, line)) { RAVELOG_ERROR("%s: %s\n", name.c_str(), impl_.last_error().c_str()); return false; }
    return true;
  }
 private:
  pocs::MCModule impl_;
};

InterfaceBasePtr CreateInterfaceValidated(InterfaceType type, const std::string& interfacename,
                                          std::istream& sinput, EnvironmentBasePtr penv) {
  if (type == PT_Module && interfacename == "mcmodule") return InterfaceBasePtr(new MCModule(penv, sinput));
  return InterfaceBasePtr();
}
void GetPluginAttributesValidated(PLUGININFO& info) { info.interfacenames[PT_Module].push_back("MCModule"); }
OPENRAVE_PLUGIN_API void DestroyPlugin() {}
