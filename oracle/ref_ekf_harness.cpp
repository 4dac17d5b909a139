// ref_ekf_harness.cpp -- the REFERENCE's EKF arithmetic, compiled from its own text.
//
// MCSimulator.h as a whole needs <openrave/plugin.h> (absent here; no stand-in is written for it), but the
// member functions that hold the estimator's scalar arithmetic use no OpenRAVE symbol: `observation`,
// `sampleObservation`, `sampleOdometry`, `prediction`, `inverseOdometry`, `generateV_EKF`, `makeHRow`,
// `generateM_EKF`, `generateG_EKF`, `generateL` (MCSimulator.h:368-553) and `EKFpredict`, `EKFupdate`
// (:868-929) read four data members only -- `landmarks`, `alphas`, `Q`, and Armadillo.  `make -C oracle ref_ekf`
// (build container only) cuts exactly those line ranges, plus the free helpers `squareNum`, `sampleNormal`,
// `angleWrap`, `roundAngle` (:43-69), out of the header WHERE IT LIES into oracle/_ref/ (git-ignored generated
// files, never committed, never copied into the repository) after checking that the ranges still begin and end
// where expected, and this file #includes them inside a class that supplies the four members.  What runs is
// the reference's own text against the vendored Armadillo -- the reference, not a restatement of it.
// Test infrastructure only (tests/test_oracle_vs_ref_ekf.py); output oracle/_ref/libpocs_ref_ekf.so.
#include <armadillo>
#include <cmath>
#include <vector>

using namespace arma;
#define Debug(x)
#include "_ref/mcsim_helpers.inc"          // MCSimulator.h:43-69

struct RefEkf {
  arma::Mat<double> alphas;                // 1 x 4, MCSimulator.h:95
  double Q;                                // :97
  arma::Mat<double> landmarks;             // 2 x L, :100
  int numLandmarks;
#include "_ref/mcsim_members_a.inc"        // :368-553
#include "_ref/mcsim_members_b.inc"        // :868-929
};

namespace {
RefEkf g;
arma::Mat<double> col3(const double* v) { arma::Mat<double> m(3, 1); for (int i = 0; i < 3; ++i) m(i, 0) = v[i]; return m; }
arma::Mat<double> mat3(const double* r) { arma::Mat<double> m(3, 3); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = r[3 * i + j]; return m; }
void out3(const arma::Mat<double>& m, double* v) { for (int i = 0; i < 3; ++i) v[i] = m(i, 0); }
void out9(const arma::Mat<double>& m, double* r) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[3 * i + j] = m(i, j); }
}  // namespace

extern "C" {

void refe_configure(const double* alphas4, double Q, const double* lx, const double* ly, int L) {
  g.alphas = arma::Mat<double>(1, 4);
  for (int i = 0; i < 4; ++i) g.alphas(0, i) = alphas4[i];
  g.Q = Q;
  g.landmarks = arma::Mat<double>(2, L);
  for (int l = 0; l < L; ++l) { g.landmarks(0, l) = lx[l]; g.landmarks(1, l) = ly[l]; }
  g.numLandmarks = L;
}
double refe_angle_wrap(double a) { return angleWrap(a); }
void refe_prediction(const double* x, const double* u, double* out) { auto a = col3(x), b = col3(u); out3(g.prediction(a, b), out); }
void refe_inverse_odometry(const double* p1, const double* p2, double* out) { auto a = col3(p1), b = col3(p2); out3(g.inverseOdometry(a, b), out); }
void refe_generate_M(const double* u, double* M9) { auto a = col3(u); out9(g.generateM_EKF(a), M9); }
void refe_generate_G(const double* mu, const double* u, double* G9) { auto a = col3(mu), b = col3(u); out9(g.generateG_EKF(a, b), G9); }
void refe_generate_V(const double* mu, const double* u, double* V9) { auto a = col3(mu), b = col3(u); out9(g.generateV_EKF(a, b), V9); }
void refe_generate_L(const double* nominal, const double* estimated, const double* goal, const double* control, double* L9) {
  auto a = col3(nominal), b = col3(estimated), c = col3(goal), d = col3(control);
  out9(g.generateL(a, b, c, d), L9);
}
// the applied control as EKF_GaussProp forms it from the gain (:714-726): nominalcontrol + gain * (estimated - nominal)
void refe_applied_control(const double* nominal, const double* estimated, const double* goal, const double* control, double* out) {
  auto a = col3(nominal), b = col3(estimated), c = col3(goal), d = col3(control);
  arma::Mat<double> gain = g.generateL(a, b, c, d);
  arma::Mat<double> statedeviation = b - a;
  arma::Mat<double> controldeviation = gain * statedeviation;
  arma::Mat<double> appliedcontrol = d + controldeviation;
  out3(appliedcontrol, out);
}
void refe_make_h_row(const double* state, int lid, double* H3) { auto a = col3(state); out3(g.makeHRow(a, lid), H3); }
double refe_observation(const double* state, int lid) { auto a = col3(state); return g.observation(a, lid); }
void refe_ekf_predict(const double* mu, const double* S9, const double* u, const double* M9, double* pmu, double* pS9) {
  auto a = col3(mu), S = mat3(S9), b = col3(u), M = mat3(M9);
  arma::Mat<double> pm, pS;
  g.EKFpredict(a, S, b, M, g.Q, pm, pS);
  out3(pm, pmu); out9(pS, pS9);
}
void refe_ekf_update(const double* pmu, const double* pS9, const double* z, int L, double* mu, double* S9) {
  auto a = col3(pmu), S = mat3(pS9);
  arma::Mat<double> meas(1, L), nm, ns;
  for (int l = 0; l < L; ++l) meas(0, l) = z[l];
  g.EKFupdate(a, S, meas, g.Q, nm, ns);
  out3(nm, mu); out9(ns, S9);
}
// sampleOdometry after arma_rng::set_seed(seed): the noisy control, the state it leads to, and the TAPE of the three
// standard normals it consumed, in its order (r1, tr, r2: :403-405)
void refe_sample_odometry(const double* state, const double* u, unsigned seed, double* noisy, double* newstate, double* tape3) {
  arma::arma_rng::set_seed(seed);
  for (int i = 0; i < 3; ++i) tape3[i] = randn();
  arma::arma_rng::set_seed(seed);
  auto a = col3(state), b = col3(u);
  arma::Mat<double> nm;
  out3(g.sampleOdometry(a, b, nm), newstate);
  out3(nm, noisy);
}
double refe_sample_observation(const double* state, int lid, unsigned seed, double* tape1) {
  arma::arma_rng::set_seed(seed);
  tape1[0] = randn();
  arma::arma_rng::set_seed(seed);
  auto a = col3(state);
  return g.sampleObservation(a, lid);
}

}  // extern "C"
