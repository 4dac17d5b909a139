// mcmodule.hpp -- C++ host-side mirror of the reference's module class on top of the C ABI.
//
// `pocs::MCModule` offers what `class MCModule : public ModuleBase` offers to OpenRAVE
// (mcsimplugin/mcsimplugin.cpp:7-232): commands registered by name with a help string
// (RegisterCommand, :13-44), each a handler `bool(std::ostream& sout, std::istream& sinput)`,
// dispatched by `SendCommand(sout, sinput)` whose first token is the command name (OpenRAVE's
// InterfaceBase::SendCommand contract).  Every handler forwards its raw token stream to
// pocs_send_command, so names, token grammar and reply text are those of the reference; a failing
// command returns false and leaves the message in last_error() instead of running into undefined
// behaviour.  Header only; link with -lpocs.  plugin/mcsimplugin_pocs.cpp wraps this class in the
// three OpenRAVE plugin entry points for a build that has OpenRAVE.
#pragma once
#include <functional>
#include <iostream>
#include <iterator>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pocs.h"

namespace pocs {

class MCModule {
 public:
  typedef std::function<bool(std::ostream&, std::istream&)> CommandFn;

  explicit MCModule(int device = 0) {
    if (pocs_create(&ctx_, device) != POCS_OK) {
      const std::string msg = pocs_last_error(ctx_);
      pocs_destroy(ctx_);
      ctx_ = nullptr;
      throw std::runtime_error("MCModule: " + msg);          // there is no CPU fallback
    }
    static const char* const kCommands[][2] = {                // mcsimplugin.cpp:13-44, same order
        {"MyCommand", "This is an example command"},
        {"ArmaCommand", "This is testing armadillo"},
        {"setAlphas", "This is to initialize alphas from python"},
        {"setQ", "This is to initialize variance of sensor noise"},
        {"setNumLandmarks", "This is to initialize number of landmark locations"},
        {"setLandmarks", "This is to initialize landmark locations"},
        {"setNumParticles", "This is to initialize number of particles for MC simulation"},
        {"setInitialCovariance", "This is to initialize first state covariance uncertainty"},
        {"setPathLength", "This is to initialize the length of the path"},
        {"setTrajectory", "This is to initialize the trajectory"},
        {"setOdometry", "This is to initialize the odometry"},
        {"runSimulation", "This is to run a MC simulation"},
        {"setNumGaussians", "This is to set number of Gaussians in mixture"},
        {"runGMMEstimation", "Use sampling-based GMM algorithm to estimate probability of collision"},
        {"setNumGMMSamples", "This is to set number of samples for GMM collision estimation"},
        {"setSeed", "(new) 64-bit seed of the counter-based random streams"},
        {"setFootprint", "(new) dx dy half_x half_y of the robot footprint"},
        {"addObstacle", "(new) cx cy half_x half_y yaw_rad of a static box"},
        {"clearObstacles", "(new) forget all obstacles"},
        {"setBatch", "(new) r independent runs advanced in lockstep per run* command"},
        {"setRunAhead", "(new) r: one run per command, the next r runs evaluated in one launch"},
    };
    for (const auto& c : kCommands) {
      const std::string name = c[0];
      RegisterCommand(name, [this, name](std::ostream& so, std::istream& si) { return Forward(name, so, si); }, c[1]);
    }
  }
  ~MCModule() { pocs_destroy(ctx_); }
  MCModule(const MCModule&) = delete;
  MCModule& operator=(const MCModule&) = delete;

  void RegisterCommand(const std::string& name, CommandFn fn, const std::string& help) {
    commands_[name] = std::make_pair(fn, help);
    order_.push_back(name);
  }

  // OpenRAVE's SendCommand: first token = command name; "help" lists the registered commands.
  bool SendCommand(std::ostream& sout, std::istream& sinput) {
    std::string name;
    if (!(sinput >> name)) { err_ = "empty command"; return false; }
    if (name == "help") {
      for (const std::string& n : order_) sout << n << " - " << commands_[n].second << "\n";
      return true;
    }
    auto it = commands_.find(name);
    if (it == commands_.end()) { err_ = "unknown command '" + name + "'"; return false; }
    return it->second.first(sout, sinput);
  }
  // convenience: the Python-side module.SendCommand("name tokens...") -> reply string
  std::string SendCommand(const std::string& line) {
    std::istringstream in(line);
    std::ostringstream out;
    if (!SendCommand(out, in)) throw std::runtime_error(err_);
    return out.str();
  }

  const std::string& last_error() const { return err_; }
  pocs_ctx* context() { return ctx_; }

  // "<name> <rest of sinput>": the line pocs_send_command takes, from a command handler's arguments --
  // `sinput` is positioned behind the command name and may hold NOTHING more: the reference's two
  // estimator commands take no tokens (mcsimplugin.cpp:66-81; MCSimulation.py:241,243 sends the bare
  // names).  The remainder is read character by character: `line << sinput.rdbuf()` would set failbit
  // on `line` for an exhausted stream (operator<<(streambuf*) inserting no character) and lose the
  // command.  No GPU in here: tests/mcmodule_demo.cpp checks it on the CPU.
  static std::string CommandLine(const std::string& name, std::istream& sinput) {
    const std::string rest((std::istreambuf_iterator<char>(sinput)), std::istreambuf_iterator<char>());
    return name + " " + rest;
  }

  // THE handler behind every registered command, and what an OpenRAVE adapter binds its commands to
  // (plugin/mcsimplugin_pocs.cpp): forwards to pocs_send_command, writes the reply to sout, false on error.
  bool Forward(const std::string& name, std::ostream& sout, std::istream& sinput) {
    const std::string line = CommandLine(name, sinput);
    std::vector<char> reply(1 << 16, '\0');
    const int rc = pocs_send_command(ctx_, line.c_str(), reply.data(), reply.size());
    if (rc != POCS_OK) { err_ = pocs_last_error(ctx_); return false; }
    sout << reply.data();
    return true;
  }

 private:

  pocs_ctx* ctx_ = nullptr;
  std::map<std::string, std::pair<CommandFn, std::string> > commands_;
  std::vector<std::string> order_;
  std::string err_;
};

}  // namespace pocs
