// ref_harness.cpp -- compiles the REFERENCE's own sources (vendored Armadillo headers and
// GM_Model.h, included by path from /root/reference, never copied) and exposes the handful of
// evaluations the hot path relies on, so tests can compare oracle/pocs_oracle.c with the real
// code.  Built by `make -C oracle ref` only where /root/reference is mounted; output goes to
// oracle/_ref/.  No LAPACK/BLAS: everything used here is header-only in Armadillo 8.400
// (chol / mvnrnd are therefore not reachable and are pinned by properties instead).
// MCSimulator.h is not included: it needs <openrave/plugin.h>, which does not exist here.
#include <armadillo>
#include <chrono>
#include <random>
#include <vector>

#define private public            // GM_Model::generator / weights / weighted_dist are private
#include "GM_Model.h"             // -I/root/reference/mcsimplugin
#undef private

extern "C" {

// arma::mean(X,1) and arma::cov(X.t()) exactly as truncateGMM calls them (MCSimulator.h:597-598).
// rows: n x 3 (x y theta per row) -> X is 3 x n.
void ref_mean_cov(const double* rows, int n, double* mean3, double* cov9) {
  arma::Mat<double> X(3, n);
  for (int c = 0; c < n; ++c) for (int r = 0; r < 3; ++r) X(r, c) = rows[3 * c + r];
  arma::Mat<double> m = arma::mean(X, 1);
  arma::Mat<double> C = arma::cov(X.t());
  for (int r = 0; r < 3; ++r) mean3[r] = m(r, 0);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cov9[3 * r + c] = C(r, c);
}

// arma::normalise(counts,1,1).row(1) as in MCSimulator.h:618-622; counts is 2 x K (row 0 =
// colliding, row 1 = free), given row-major.
void ref_normalise_rows(const double* counts_2xK, int K, double* weights) {
  arma::Mat<double> Cn(2, K);
  for (int r = 0; r < 2; ++r) for (int k = 0; k < K; ++k) Cn(r, k) = counts_2xK[r * K + k];
  arma::Mat<double> W = arma::normalise(Cn, 1, 1);
  for (int k = 0; k < K; ++k) weights[k] = W(1, k);
}

// A * B * C.t() and A * B * A.t() + R as EKFpredict evaluates them (MCSimulator.h:874,878); 3x3 row-major.
void ref_vmvt(const double* V, const double* M, double* out) {
  arma::Mat<double> v(3, 3), m(3, 3);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { v(r, c) = V[3 * r + c]; m(r, c) = M[3 * r + c]; }
  arma::Mat<double> R = v * m * v.t();
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out[3 * r + c] = R(r, c);
}
void ref_gsgt_plus_r(const double* G, const double* S, const double* R, double* out) {
  arma::Mat<double> g(3, 3), s(3, 3), rr(3, 3);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { g(r, c) = G[3 * r + c]; s(r, c) = S[3 * r + c]; rr(r, c) = R[3 * r + c]; }
  arma::Mat<double> P = g * s * g.t() + rr;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out[3 * r + c] = P(r, c);
}

// One scalar measurement update written with the same Armadillo expressions as EKFupdate
// (MCSimulator.h:896-921): H (1x3 row), S = H P H^T + Q, K = P H^T S.i(), mu += K*innov,
// P = (I - K H) P.
void ref_scalar_update(const double* H3, double Q, double innov, double* mu3, double* P9) {
  arma::Mat<double> H(1, 3), P(3, 3), mu(3, 1);
  for (int c = 0; c < 3; ++c) { H(0, c) = H3[c]; mu(c, 0) = mu3[c]; }
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) P(r, c) = P9[3 * r + c];
  arma::Mat<double> S = H * P * H.t() + Q;
  arma::Mat<double> K = P * H.t() * S.i();
  mu = mu + K * (innov);
  P = (arma::eye<arma::Mat<double> >(3, 3) - K * H) * P;
  for (int c = 0; c < 3; ++c) mu3[c] = mu(c, 0);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) P9[3 * r + c] = P(r, c);
}

// arma::prod(1 - probabilities, 1) and 1 - that, MCSimulator.h:848-856.
double ref_final_combine(const double* probs, int W) {
  arma::Mat<double> p(1, W);
  for (int i = 0; i < W; ++i) p(0, i) = probs[i];
  arma::Mat<double> freeMat = 1 - p;
  arma::Mat<double> pr = arma::prod(freeMat, 1);
  return 1 - pr(0, 0);
}

// GM_Model's component selection (GM_Model.h:89-93,119-124): N draws of its
// std::discrete_distribution with the engine seeded explicitly -> counts per component.
// (sampleNPoints itself then calls mvnrnd, which needs LAPACK; only the split is exercised.)
void ref_gm_model_counts(const double* weights, int K, int N, unsigned seed, int* counts) {
  GM_Model gm;
  gm.numGaussians = K;
  std::vector<double> w(weights, weights + K);
  gm.updateWeights(w);
  gm.generator.seed(seed);
  for (int k = 0; k < K; ++k) counts[k] = 0;
  for (int i = 0; i < N; ++i) ++counts[gm.weighted_dist(gm.generator)];
}
}
