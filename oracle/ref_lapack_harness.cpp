// ref_lapack_harness.cpp -- the REFERENCE's sampler as it really runs: the vendored Armadillo
// headers + GM_Model.h compiled where they lie (included by path from /root/reference, never
// copied) WITH LAPACK/BLAS, so that arma::chol / arma::mvnrnd / GM_Model::sampleNPoints execute
// their own code (glue_mvnrnd_meat.hpp:92-147 -> op_chol -> LAPACK potrf, auxlib_meat.hpp:1760-1797;
// the eigen fallback :100-132 -> dsyevd).  LAPACK/BLAS come from the OpenBLAS that ships inside the
// image's scipy wheel (scipy.libs/libscipy_openblas-*.so: a library that is present, LP64, symbols
// prefixed `scipy_`, mapped with -D<sym>_=scipy_<sym>_ by the Makefile).  Built by
// `make -C oracle ref_lapack` only where /root/reference is mounted; output in oracle/_ref/
// (git-ignored; the built library travels to the GPU box, no source does).  Test infrastructure only.
//
// Armadillo's generator cannot be shared with the build (it is mt19937_64 + std::normal_distribution,
// arma_rng_cxx11.hpp:24-111), so every entry point also returns the TAPE of standard normals the
// reference consumed: re-seed identically, replay randn<mat>(3, n) in the reference's order
// (column-major fill, arma_rng.hpp:369-432).  The restatement fed with that tape must reproduce the
// reference's points.
#include <armadillo>
#include <chrono>
#include <random>
#include <sstream>
#include <vector>

#define private public            // GM_Model::generator / weighted_dist are private (GM_Model.h:42-50)
#include "GM_Model.h"             // -I/root/reference/mcsimplugin
#undef private

namespace {
arma::Mat<double> mat3(const double* row_major9) {
  arma::Mat<double> C(3, 3);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = row_major9[3 * r + c];
  return C;
}
struct Quiet {                    // GM_Model prints through std::cout (GMMDebug is on, GM_Model.h:9-15)
  std::streambuf* old;
  std::ostringstream sink;
  Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {

// arma::chol(C, "lower") -> row-major 3x3; returns 0 when Armadillo reports failure.
int refl_chol_lower(const double* cov9, double* L9) {
  arma::Mat<double> D;
  if (!arma::chol(D, mat3(cov9), "lower")) return 0;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L9[3 * r + c] = D(r, c);
  return 1;
}

// arma::mvnrnd(out, M, C, n) after arma_rng::set_seed(seed).  points / tape: 3 x n column-major
// (x, y, theta triples), exactly the layout of the reference's matrices.  Returns mvnrnd's status.
int refl_mvnrnd(const double* mean3, const double* cov9, int n, unsigned seed, double* points, double* tape) {
  arma::Mat<double> M(3, 1);
  for (int r = 0; r < 3; ++r) M(r, 0) = mean3[r];
  const arma::Mat<double> C = mat3(cov9);
  arma::arma_rng::set_seed(seed);
  const arma::Mat<double> Z = arma::randn<arma::Mat<double> >(3, n);      // what mvnrnd will draw
  arma::arma_rng::set_seed(seed);
  arma::Mat<double> X;
  const bool ok = arma::mvnrnd(X, M, C, (arma::uword)n);
  if (!ok) return 0;
  for (int i = 0; i < 3 * n; ++i) { points[i] = X.memptr()[i]; tape[i] = Z.memptr()[i]; }
  return 1;
}

// GM_Model::sampleNPoints (GM_Model.h:83-116) on a mixture given as K means (3 each), K row-major
// covariances, K weights; both generators seeded (Armadillo's, and the model's
// std::default_random_engine, GM_Model.h:53-54).  counts[K]; points / tape: 3 x N column-major, the
// components' blocks one after the other as sampleNPoints produces them.  Returns 1.
int refl_sample_n_points(const double* means, const double* covs, const double* weights, int K, int N,
                         unsigned arma_seed, unsigned gen_seed, int* counts, double* points, double* tape) {
  Quiet q;
  GM_Model gm;
  arma::Mat<double> m0(3, 1, arma::fill::zeros);
  gm.initModel(K, m0, arma::eye<arma::Mat<double> >(3, 3));
  for (int k = 0; k < K; ++k) {
    for (int r = 0; r < 3; ++r) gm.means[k](r, 0) = means[3 * k + r];
    gm.covariances[k] = mat3(covs + 9 * k);
  }
  std::vector<double> w(weights, weights + K);
  gm.updateWeights(w);
  gm.generator.seed(gen_seed);
  arma::arma_rng::set_seed(arma_seed);
  std::vector<arma::Mat<double> > pts;
  gm.sampleNPoints(N, pts);
  size_t off = 0;
  for (int k = 0; k < K; ++k) {
    counts[k] = (int)pts[k].n_cols;
    for (size_t i = 0; i < (size_t)pts[k].n_elem; ++i) points[off + i] = pts[k].memptr()[i];
    off += pts[k].n_elem;
  }
  arma::arma_rng::set_seed(arma_seed);                                     // replay the draws, same order
  off = 0;
  for (int k = 0; k < K; ++k) {
    const arma::Mat<double> Z = arma::randn<arma::Mat<double> >(3, counts[k]);
    for (size_t i = 0; i < (size_t)Z.n_elem; ++i) tape[off + i] = Z.memptr()[i];
    off += Z.n_elem;
  }
  return 1;
}
}
