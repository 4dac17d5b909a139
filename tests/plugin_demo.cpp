// plugin_demo.cpp -- compiles plugin/mcsimplugin_pocs.cpp itself (against tests/openrave_shim, a test double)
// and drives it the way OpenRAVE and MCSimulation.py do: GetPluginAttributesValidated, then
// CreateInterfaceValidated(PT_Module, "mcmodule") on an environment holding the boxes of pr2test2.env.xml and a
// robot, then every command of the reference through InterfaceBase::SendCommand(sout, sinput) -- the two
// estimator commands as the bare names MCSimulation.py:241,243 sends.
//   usage: plugin_demo --cmdline-only            (no GPU: the command-line assembly only)
//          plugin_demo <plan.txt> <N> <K> <seed>
#include "../plugin/mcsimplugin_pocs.cpp"

#include <cmath>
#include <fstream>
#include <iomanip>

static std::string fmt(double v) { std::ostringstream o; o << std::setprecision(17) << v; return o.str(); }

static KinBodyPtr box_body(const std::string& name, double cx, double cy, double cz, double hx, double hy, double hz, double yaw = 0.0) {
  KinBodyPtr b(new KinBody);
  b->name = name;
  KinBody::LinkPtr l(new KinBody::Link);
  l->name = "base";
  l->t.trans = Vector(cx, cy, cz);                         // the body's pose carries the position,
  KinBody::Link::GeometryPtr g(new KinBody::Link::Geometry);
  g->t.R[0] = std::cos(yaw); g->t.R[1] = -std::sin(yaw); g->t.R[3] = std::sin(yaw); g->t.R[4] = std::cos(yaw);   // the geometry its yaw
  g->extents = Vector(hx, hy, hz);
  l->geoms.push_back(g);
  b->links.push_back(l);
  return b;
}

static bool send(InterfaceBasePtr m, const std::string& line, std::string* reply = nullptr) {
  std::istringstream in(line);
  std::ostringstream out;
  const bool ok = m->SendCommand(out, in);
  if (reply) *reply = out.str();
  return ok;
}

int main(int argc, char** argv) {
  if (argc == 2 && std::string(argv[1]) == "--cmdline-only") {
    // what a handler receives for "runGMMEstimation": the stream behind the name, exhausted
    std::istringstream in("runGMMEstimation");
    std::string name; in >> name;
    const std::string line = pocs::MCModule::CommandLine(name, in);
    std::istringstream in2("setQ 0.04"); in2 >> name;
    const std::string line2 = pocs::MCModule::CommandLine(name, in2);
    // the idiom round 2's adapter used, for the record: inserting an exhausted streambuf sets failbit
    std::istringstream in3("runGMMEstimation"); in3 >> name;
    std::stringstream old; old << name << ' ' << in3.rdbuf();
    std::string again; const bool old_ok = static_cast<bool>(old >> again);
    std::printf("LINE [%s]\nLINE2 [%s]\nOLD_IDIOM_OK %d\n", line.c_str(), line2.c_str(), old_ok ? 1 : 0);
    return 0;
  }
  if (argc != 5) { std::fprintf(stderr, "usage\n"); return 2; }
  std::ifstream pf(argv[1]);
  std::string line;
  std::vector<double> v;
  int W = -1;
  while (std::getline(pf, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    if (W < 0) { is >> W; continue; }
    double a, b, c; is >> a >> b >> c; v.push_back(a); v.push_back(b); v.push_back(c);
  }
  // pr2test2.env.xml as OpenRAVE would hold it: floor, four side walls, the middle wall's two pieces and its
  // lintel, the small box, a cylinder the adapter cannot represent -- and the robot
  EnvironmentBasePtr env(new EnvironmentBase);
  env->bodies.push_back(box_body("floor", 0, 0, -0.005, 4.0, 2.0, 0.005));
  env->bodies.push_back(box_body("SideWall1", 3.9, 0.0, 0.1, 0.1, 1.8, 0.1));
  env->bodies.push_back(box_body("SideWall2", -3.9, 0.0, 0.1, 0.1, 1.8, 0.1));
  env->bodies.push_back(box_body("SideWall3", 0.0, 1.9, 0.1, 4.0, 0.1, 0.1));
  env->bodies.push_back(box_body("SideWall4", 0.0, -1.9, 0.1, 4.0, 0.1, 0.1));
  env->bodies.push_back(box_body("MidWallLow", 0.8, -0.565, 1.0, 0.1, 1.235, 1.0));
  env->bodies.push_back(box_body("MidWallHigh", 0.8, 1.65, 1.0, 0.1, 0.15, 1.0));
  env->bodies.push_back(box_body("Lintel", 0.8, 1.085, 2.25, 0.1, 0.415, 0.25));
  env->bodies.push_back(box_body("TibitsBox1", 3.5, -1.3, 0.806, 0.025, 0.0935, 0.066));
  env->bodies.back()->links[0]->geoms.push_back(KinBody::Link::GeometryPtr(new KinBody::Link::Geometry));
  env->bodies.back()->links[0]->geoms.back()->type = GT_Cylinder;
  RobotBasePtr robot(new RobotBase);
  robot->name = "PR2";
  robot->links.push_back(KinBody::LinkPtr(new KinBody::Link));
  robot->links[0]->name = "base_link";
  robot->links[0]->local.pos = Vector(0, 0, 0.2);
  robot->links[0]->local.extents = Vector(0.334, 0.334, 0.2);
  env->bodies.push_back(robot);
  env->robots.push_back(robot);
  try {
    PLUGININFO info;
    GetPluginAttributesValidated(info);
    std::printf("ADVERTISED %s\n", info.interfacenames[PT_Module].at(0).c_str());
    std::istringstream none("");
    if (CreateInterfaceValidated(PT_Planner, "mcmodule", none, env) || CreateInterfaceValidated(PT_Module, "other", none, env)) { std::puts("WRONG_INTERFACE"); return 1; }
    InterfaceBasePtr m = CreateInterfaceValidated(PT_Module, "mcmodule", none, env);      // lower-case, mcsimplugin.cpp:237
    if (!m) { std::puts("NO_MODULE"); return 1; }
    int ok = 0, sent = 0;
    std::string r;
    auto S = [&](const std::string& l) { ++sent; if (send(m, l, &r)) ++ok; else std::fprintf(stderr, "failed: %s\n", l.substr(0, 40).c_str()); return r; };
    std::printf("MYCOMMAND %s\n", S("MyCommand").c_str());
    S("ArmaCommand");
    S("setAlphas 6.25e-08 6.25e-06 6.25e-06 6.25e-06 ");
    S("setQ 0.04000000000000001");
    S("setNumLandmarks 8");
    S("setLandmarks 3 -3 0 0 -3 3 -3 3 0 0 2 -2 2 2 -2 -2 ");
    S(std::string("setNumParticles ") + argv[2]);
    S("setInitialCovariance 0.001 0 0 0 0.001 0 0 0 0.001 ");
    S("setPathLength " + std::to_string(W));
    std::string traj = "setTrajectory ", odom = "setOdometry ";
    for (int c = 0; c < 3; ++c) for (int i = 0; i < W; ++i) traj += fmt(v[3 * i + c]) + " ";
    for (int c = 0; c < 3; ++c) for (int i = 0; i < W - 1; ++i) odom += fmt(v[3 * (W + i) + c]) + " ";
    S(traj);
    S(odom);
    S(std::string("setNumGaussians ") + argv[3]);
    S(std::string("setNumGMMSamples ") + argv[2]);
    S(std::string("setSeed ") + argv[4]);
    std::printf("GMM %s\n", S("runGMMEstimation").c_str());                // the bare name: MCSimulation.py:243
    std::printf("GMM2 %s\n", S("runGMMEstimation").c_str());               // the next run, served from the run-ahead launch
    S(std::string("setSeed ") + argv[4]);
    std::printf("MC %s\n", S("runSimulation").c_str());                    // MCSimulation.py:241
    std::printf("COMMANDS %d of %d\n", ok, sent);
    std::printf("BADCMD %d\n", send(m, "setLandmarks 1 2 3") ? 1 : 0);
    std::printf("UNKNOWN %d\n", send(m, "noSuchCommand") ? 1 : 0);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
