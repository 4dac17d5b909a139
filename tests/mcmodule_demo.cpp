// mcmodule_demo.cpp -- test program for csrc/mcmodule.hpp: replays the command sequence of
// MCSimulation.py:154-207,238-245 through pocs::MCModule::SendCommand(ostream, istream) and prints
// "GMM <p>\nMC <p>".  usage: mcmodule_demo <plan.txt> <env.txt> <N> <K> <seed> | mcmodule_demo --help-only
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <limits>
#include "../probability-of-collision-for-safe-planning_amd/csrc/mcmodule.hpp"

static std::string fmt(double v) { std::ostringstream o; o << std::setprecision(17) << v; return o.str(); }

int main(int argc, char** argv) {
  if (argc == 2 && std::string(argv[1]) == "--help-only") { std::puts("built"); return 0; }
  if (argc != 6) { std::fprintf(stderr, "usage\n"); return 2; }
  std::ifstream pf(argv[1]), ef(argv[2]);
  std::string line;
  std::vector<double> v;
  int W = -1;
  while (std::getline(pf, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    if (W < 0) { is >> W; continue; }
    double a, b, c; is >> a >> b >> c; v.push_back(a); v.push_back(b); v.push_back(c);
  }
  try {
    pocs::MCModule mod(0);
    std::ostringstream help; std::istringstream hi("help");
    mod.SendCommand(help, hi);
    mod.SendCommand("clearObstacles");
    while (std::getline(ef, line)) {
      std::istringstream is(line); std::string kind; is >> kind;
      std::string rest; std::getline(is, rest);
      if (kind == "footprint") mod.SendCommand("setFootprint" + rest);
      if (kind == "box") mod.SendCommand("addObstacle" + rest);
    }
    mod.SendCommand("setAlphas 6.25e-08 6.25e-06 6.25e-06 6.25e-06 ");
    mod.SendCommand("setQ 0.04000000000000001");
    mod.SendCommand("setNumLandmarks 8");
    mod.SendCommand("setLandmarks 3 -3 0 0 -3 3 -3 3 0 0 2 -2 2 2 -2 -2 ");
    mod.SendCommand(std::string("setNumParticles ") + argv[3]);
    mod.SendCommand("setInitialCovariance 0.001 0 0 0 0.001 0 0 0 0.001 ");
    mod.SendCommand("setPathLength " + std::to_string(W));
    std::string traj = "setTrajectory ", odom = "setOdometry ";
    for (int c = 0; c < 3; ++c) for (int i = 0; i < W; ++i) traj += fmt(v[3 * i + c]) + " ";
    for (int c = 0; c < 3; ++c) for (int i = 0; i < W - 1; ++i) odom += fmt(v[3 * (W + i) + c]) + " ";
    mod.SendCommand(traj);
    mod.SendCommand(odom);
    mod.SendCommand(std::string("setNumGaussians ") + argv[4]);
    mod.SendCommand(std::string("setNumGMMSamples ") + argv[3]);
    mod.SendCommand(std::string("setSeed ") + argv[5]);
    std::printf("GMM %s\n", mod.SendCommand("runGMMEstimation").c_str());
    mod.SendCommand(std::string("setSeed ") + argv[5]);
    std::printf("MC %s\n", mod.SendCommand("runSimulation").c_str());
    // one run per command with and without run-ahead: the same six MC results
    std::string plain, ahead;
    mod.SendCommand(std::string("setSeed ") + argv[5]);
    for (int i = 0; i < 6; ++i) plain += mod.SendCommand("runSimulation") + " ";
    mod.SendCommand("setRunAhead 4");
    mod.SendCommand(std::string("setSeed ") + argv[5]);
    for (int i = 0; i < 6; ++i) ahead += mod.SendCommand("runSimulation") + " ";
    mod.SendCommand("setRunAhead 1");
    std::printf("AHEAD %d\n", plain == ahead ? 1 : 0);
    std::istringstream bad("setLandmarks 1 2 3"); std::ostringstream sink;
    std::printf("BADCMD %d\n", mod.SendCommand(sink, bad) ? 1 : 0);
    const std::string helptext = help.str();
    std::printf("HELPLINES %d\n", (int)std::count(helptext.begin(), helptext.end(), '\n'));
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
