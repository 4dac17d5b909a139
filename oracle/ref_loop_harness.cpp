// ref_loop_harness.cpp -- the REFERENCE's estimator loop, compiled from its own text.
//
// MCSimulator.h as a whole needs <openrave/plugin.h> (absent here; no stand-in is written for it).  Exactly one
// of its member functions touches OpenRAVE: `checkCollision(const config&)` (:269-285: SetActiveDOFValues +
// CheckCollision on the live scene).  Everything else of the class -- the data members (:94-129), the setters
// (:158-235), `checkMatrixCollisions` and the Armadillo overload of `checkCollision` (:241-266), the particle
// functions, `initGMM`, `runGMMEstimation`, `runSimulation` (:287-365), the EKF arithmetic (:368-553, :868-929) and
// the loop itself, `GMM_GaussProp` / `truncateGMM` / `EKF_GaussProp` (:559-864) -- uses Armadillo and GM_Model.h
// only.  `make -C oracle ref_loop` (build container only) cuts those line ranges, and the free helpers of :43-91,
// out of the header WHERE IT LIES into oracle/_ref/*.inc (generated, git-ignored, deleted after the compile,
// never committed or copied into the repository) after checking that each range still begins and ends where
// expected; this file #includes them inside a struct that adds the one missing function: a `checkCollision(const
// config&)` that asks a caller-supplied 2-D predicate (the tests pass the oracle's `orc_collides` and its world).
// That is this build's documented replacement for the OpenRAVE/ODE scene, not a model of
// OpenRAVE.  What runs is the reference's own loop -- order of calls, what is predicted from what, which control
// feeds which noise, the truncation and the final product -- against the vendored Armadillo.
//
// Random numbers: the loop draws through arma::randn (std::rand underneath in this configuration, arma_rng_cxx98)
// and, for the mixture's component choice, GM_Model's private std::default_random_engine.  refl2_record_* replay
// the SAME sequence of arma calls (same shapes, same order) after the same seed and hand back the normals as a
// tape, so that the oracle can be run on the reference's own noise and compared value for value.
// Test infrastructure only (tests/test_oracle_vs_ref_loop.py); output oracle/_ref/libpocs_ref_loop.so.
#include <armadillo>
#include <cassert>
#include <cmath>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#define private public            // GM_Model::generator is private (GM_Model.h:42-50); seeded below for repeatable runs
#include "GM_Model.h"             // -I/root/reference/mcsimplugin
#undef private

using namespace arma;
#define Debug(x)
#define Debug2(x)
#define Debug3(x)
#define Debug4(x)
#include "_ref/mcsim_loop_helpers.inc"      // MCSimulator.h:43-91

typedef int (*collide_fn)(double x, double y, double th, const void* world);

struct RefLoop {
#include "_ref/mcsim_loop_members.inc"      // :94-129   data members
#include "_ref/mcsim_loop_setters.inc"      // :158-235  setTrajectory .. setQ
#include "_ref/mcsim_loop_collide.inc"      // :241-266  checkMatrixCollisions, checkCollision(arma::Mat<double>&)
  // the one OpenRAVE-bound member (:269-285), replaced: the caller's 2-D predicate; every checked pose is counted
  collide_fn collide = nullptr;
  const void* world = nullptr;
  long long checked = 0;
  bool checkCollision(const config& c) { ++checked; return collide(c[0], c[1], c[2], world) != 0; }
#include "_ref/mcsim_loop_particles.inc"    // :287-365  initParticles .. runSimulation
#include "_ref/mcsim_loop_ekf_a.inc"        // :368-553
#include "_ref/mcsim_loop_main.inc"         // :559-864  GMM_GaussProp, truncateGMM, EKF_GaussProp
#include "_ref/mcsim_loop_ekf_b.inc"        // :868-929
};

namespace {
RefLoop* g = nullptr;
std::string g_text;               // what the last run printed
std::string g_error;              // what() of the exception that ended the last run, if one did
struct Quiet {                    // the loop prints whole matrices through std::cout
  std::streambuf* old;
  std::ostringstream sink;
  Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { std::cout.rdbuf(old); g_text = sink.str(); }
};
arma::Mat<double> by_component(const double* v, int rows, int cols) {        // v[r * cols + c]
  arma::Mat<double> m(rows, cols);
  for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) m(r, c) = v[(size_t)r * cols + c];
  return m;
}
}  // namespace

extern "C" {

// traj: 3 x W by component, odom: 3 x (W-1) by component (mcsimplugin.cpp:83-113's layout), cov0 row-major.
void refl2_configure(const double* alphas4, double Q, const double* lx, const double* ly, int L, const double* traj,
                     const double* odom, int W, const double* cov0, int particles, int gaussians, int samples,
                     collide_fn fn, const void* world) {
  Quiet q;
  delete g;
  g = new RefLoop();
  g->alphas = ones<arma::Mat<double> >(1, 4);                 // the constructor's line (:143)
  g->setAlphas(std::vector<double>(alphas4, alphas4 + 4));
  g->setQ(Q);
  arma::Mat<double> lm(2, L);
  for (int l = 0; l < L; ++l) { lm(0, l) = lx[l]; lm(1, l) = ly[l]; }
  g->setLandmarks(lm);
  g->setNumLandmarks(L);
  g->setTrajectory(by_component(traj, 3, W));
  g->setOdometry(by_component(odom, 3, W - 1));
  g->setPathLength(W);
  g->setInitialCovariance(by_component(cov0, 3, 3));
  g->setNumParticles(particles);
  g->setNumGaussians(gaussians);
  g->setNumGMMSamples(samples);
  g->collide = fn;
  g->world = world;
}

// runSimulation() after arma_rng::set_seed(seed).  Out: the belief after the last step (mu 3, cov 9 row-major), the
// particles (3 x N column-major = x y theta triples) and their collision counts.  Returns the proportion.
double refl2_run_mc(unsigned seed, double* mu, double* cov, double* particles, unsigned* hits, long long* checked) {
  Quiet q;
  g_error.clear();
  arma::arma_rng::set_seed(seed);
  g->checked = 0;
  double p;
  try { p = g->runSimulation(); }
  catch (const std::exception& e) { g_error = e.what(); return std::nan(""); }   // Armadillo throws (std::logic_error / runtime_error)
  for (int i = 0; i < 3; ++i) mu[i] = g->mu(i, 0);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cov[3 * r + c] = g->cov(r, c);
  for (size_t i = 0; i < (size_t)g->mcparticles.n_elem; ++i) particles[i] = g->mcparticles.memptr()[i];
  for (size_t i = 0; i < (size_t)g->particlecollisions.n_elem; ++i) hits[i] = g->particlecollisions.memptr()[i];
  *checked = g->checked;
  return p;
}
// The normals runSimulation() draws after that seed, in its order and with its call shapes: randn(3, N) once
// (initParticles' mvnrnd, :290), then per step three scalar randn() (sampleOdometry :403-405) and L more
// (sampleObservation per landmark, :786-789).  init: 3 x N; chain: (W-1) x (3 + L).
void refl2_record_mc(unsigned seed, int N, int W, int L, double* init, double* chain) {
  arma::arma_rng::set_seed(seed);
  const arma::Mat<double> Z = arma::randn<arma::Mat<double> >(3, N);
  for (size_t i = 0; i < (size_t)Z.n_elem; ++i) init[i] = Z.memptr()[i];
  for (int s = 0; s < W - 1; ++s) for (int k = 0; k < 3 + L; ++k) chain[(size_t)s * (3 + L) + k] = randn();
}

// runGMMEstimation() after both generators are seeded.  Out: the belief after the last step, the mixture after the
// last truncation (K means, K row-major covariances; the weights are private to GM_Model and stay so).
double refl2_run_gmm(unsigned seed, unsigned gen_seed, double* mu, double* cov, double* means, double* covs, long long* checked) {
  Quiet q;
  g_error.clear();
  arma::arma_rng::set_seed(seed);
  g->gmm.generator.seed(gen_seed);
  g->checked = 0;
  double p;
  try { p = g->runGMMEstimation(); }
  catch (const std::exception& e) { g_error = e.what(); return std::nan(""); }   // e.g. mvnrnd on a covariance that is not positive
                                                                                  // semi-definite, mean() of no samples: the reference dies there
  for (int i = 0; i < 3; ++i) mu[i] = g->mu(i, 0);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cov[3 * r + c] = g->cov(r, c);
  for (int k = 0; k < g->numGaussians; ++k) {       // (a Gaussian whose last truncation left no sample holds EMPTY matrices: NaN)
    const bool ok = g->gmm.means[k].n_elem == 3 && g->gmm.covariances[k].n_elem == 9;
    for (int i = 0; i < 3; ++i) means[3 * k + i] = ok ? g->gmm.means[k](i, 0) : std::nan("");
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) covs[9 * k + 3 * r + c] = ok ? g->gmm.covariances[k](r, c) : std::nan("");
  }
  *checked = g->checked;
  return p;
}
// The normals runGMMEstimation() draws: per waypoint one randn(3, counts[w][k]) per component in component order
// (sampleNPoints' mvnrnd calls, GM_Model.h:100-107), and between waypoints the chain's 3 + L scalars.  The call
// shapes depend on the component counts, which the run itself prints ("Counts Vector", GM_Model.h:95-96): the
// caller reads them from refl2_last_text and passes them in.  gmm: W x (3 x N column-major, the components' blocks one
// after the other); chain: (W-1) x (3 + L).
void refl2_record_gmm(unsigned seed, int N, int W, int L, int K, const long long* counts, double* gmm, double* chain) {
  arma::arma_rng::set_seed(seed);
  for (int w = 0; w < W; ++w) {
    size_t off = (size_t)w * 3 * N;
    for (int k = 0; k < K; ++k) {
      const arma::Mat<double> Z = arma::randn<arma::Mat<double> >(3, (arma::uword)counts[(size_t)w * K + k]);
      for (size_t i = 0; i < (size_t)Z.n_elem; ++i) gmm[off + i] = Z.memptr()[i];
      off += Z.n_elem;
    }
    if (w < W - 1) for (int k = 0; k < 3 + L; ++k) chain[(size_t)w * (3 + L) + k] = randn();
  }
}

// what() of the exception that ended the last run (NaN returned), or "".
long long refl2_last_error(char* buf, long long cap) {
  const long long n = (long long)g_error.size();
  if (buf && cap > 0) { const long long m = n < cap - 1 ? n : cap - 1; memcpy(buf, g_error.data(), (size_t)m); buf[m] = 0; }
  return n;
}

// The text the last run printed (the reference reports the per-waypoint probabilities only there, :845-846).
long long refl2_last_text(char* buf, long long cap) {
  const long long n = (long long)g_text.size();
  if (buf && cap > 0) { const long long m = n < cap - 1 ? n : cap - 1; memcpy(buf, g_text.data(), (size_t)m); buf[m] = 0; }
  return n;
}

}  // extern "C"
