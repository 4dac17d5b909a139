/* pocs_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker the GPU path is compared against.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it; the product (libpocs.so) never
 * links or includes anything under oracle/.
 *
 * What it restates (paths under /root/reference/mcsimplugin/ unless noted), in plain C99, scalar,
 * one sample at a time, following the reference's own data flow (grouped samples, explicit free
 * sets, Armadillo's cov/mean/normalise formulas):
 *   MCSimulator.h:56-69     angleWrap / roundAngle              -> orc_wrap_angle
 *   MCSimulator.h:413-431   prediction                          -> orc_prediction
 *   MCSimulator.h:434-449   inverseOdometry                     -> orc_inverse_odometry
 *   MCSimulator.h:495-513   generateM_EKF                       -> orc_generate_M
 *   MCSimulator.h:453-468,517-529,868-881  V, G, EKFpredict     -> orc_ekf_predict
 *   MCSimulator.h:368-381,470-492,883-929  observation, makeHRow, EKFupdate -> orc_ekf_update
 *   MCSimulator.h:532-553,714-726          generateL + applied control      -> inside orc_host_chain
 *   MCSimulator.h:391-410,383-387          sampleOdometry, sampleObservation -> inside orc_host_chain
 *   MCSimulator.h:287-347   initParticles, moveParticles, checkParticleCollisions,
 *                           getCollisionProportion              -> orc_run_mc
 *   GM_Model.h:57-124       initModel, sampleNPoints, updateWeights
 *   MCSimulator.h:570-642   truncateGMM (+ checkMatrixCollisions :241-253) -> orc_gmm_waypoint
 *   MCSimulator.h:649-864   EKF_GaussProp driver loop, final 1 - prod(1 - p_i) -> orc_run_gmm
 *   armadillo_bits/op_cov_meat.hpp:26-53, op_mean_meat.hpp:104-133,
 *   op_normalise_meat.hpp:85-121, glue_mvnrnd_meat.hpp:92-147   -> orc_cov_mean, orc_normalise_l1, orc_chol3_lower
 *
 * What is NOT the reference's and is the build's own specification, restated
 * here independently of the product sources:
 *   - random streams: Philox4x32-10 + Box-Muller with the log / sincos kernels below (the
 *     reference seeds Armadillo and std::default_random_engine from the clock, so it has no
 *     reproducible stream, MCSimulator.h:141, GM_Model.h:53-54);
 *   - the collision predicate: oriented footprint box vs oriented static boxes (the reference
 *     calls OpenRAVE's CheckCollision, MCSimulator.h:279, which is not in its tree);
 *   - degenerate truncation cases (a component with < 2 survivors is retired).
 *
 * Pinning status: see oracle/README.md -- pinned against the reference's fixtures where they
 * exist (odometry.dat == inverseOdometry(trajectory.dat); Armadillo's fn_cov known answers), against the
 * compiled Armadillo / GM_Model.h pieces and -- round 3 -- against the reference's own EKF member functions
 * (MCSimulator.h:368-553,868-929 cut from the header and compiled: oracle/_ref) and its whole estimator loop
 * (runSimulation / runGMMEstimation, :94-129,158-235,241-266,287-365,559-864 compiled the same way, run on seeded
 * generators; this file then runs on the SAME normals through orc_set_tapes: tests/test_oracle_vs_ref_loop.py).
 * "Parity unpinned": the OpenRAVE collision result (checkCollision(const config&), :269-285; not in the
 * reference tree -- MCSimulator.h as a whole needs <openrave/plugin.h>, absent here).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fma() is always explicit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_K 8
#define ORC_MAX_L 32
#define ORC_NMOM 11
#define ORC_STATE 16

/* ------------------------------------------------------------------------------------------ */
/* random streams (build spec "POCS numerics v9")                          */
/* ------------------------------------------------------------------------------------------ */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < rounds; ++round) {
    uint64_t prod0 = (uint64_t)c[0] * 0xD2511F53ull;
    uint64_t prod1 = (uint64_t)c[2] * 0xCD9E8D57ull;
    uint32_t hi0 = (uint32_t)(prod0 >> 32), lo0 = (uint32_t)prod0;
    uint32_t hi1 = (uint32_t)(prod1 >> 32), lo1 = (uint32_t)prod1;
    uint32_t t0 = hi1 ^ c[1] ^ k0;
    uint32_t t2 = hi0 ^ c[3] ^ k1;
    c[0] = t0; c[1] = lo1; c[2] = t2; c[3] = lo0;
    k0 += 0x9E3779B9u;  /* golden ratio */
    k1 += 0xBB67AE85u;  /* sqrt(3)-1 */
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32(ctr, key, 10, out); }

static void draw(uint64_t seed, uint64_t index, uint32_t waypoint, uint32_t stream, uint32_t slot,
                 uint32_t out[4]) {
  uint32_t ctr[4], key[2];
  ctr[0] = (uint32_t)(index & 0xffffffffu);
  ctr[1] = (uint32_t)(index >> 32);
  ctr[2] = waypoint;
  ctr[3] = (stream << 16) | slot;
  key[0] = (uint32_t)(seed & 0xffffffffu);
  key[1] = (uint32_t)(seed >> 32);
  /* 10 rounds (Random123's default) everywhere but the mixture samples (stream 3): 7, the generator's
   * Crush-resistant minimum (Salmon et al., SC'11, table 2) -- build spec, DESIGN.md section 4 */
  orc_philox4x32(ctr, key, stream == 3 ? 7 : 10, out);
}

/* natural log of a normal positive double: x = 2^k (1+f), sqrt(.5) < 1+f <= sqrt(2) */
double orc_log(double x) {
  static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  static const double G1 = 6.666666666666735130e-01, G2 = 3.999999999940941908e-01,
                      G3 = 2.857142874366239149e-01, G4 = 2.222219843214978396e-01,
                      G5 = 1.818357216161805012e-01, G6 = 1.531383769920937332e-01,
                      G7 = 1.479819860511658591e-01;
  uint64_t bits;
  memcpy(&bits, &x, 8);
  uint32_t high = (uint32_t)(bits >> 32);
  int k = (int)(high >> 20) - 1023;
  high &= 0x000fffffu;
  uint32_t bump = (high + 0x95f64u) & 0x100000u;
  high |= (bump ^ 0x3ff00000u);
  k += (int)(bump >> 20);
  bits = ((uint64_t)high << 32) | (bits & 0xffffffffull);
  double m;
  memcpy(&m, &bits, 8);
  double f = m - 1.0;
  double dk = (double)k;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double even = w * fma(w, fma(w, G6, G4), G2);
  double odd = z * fma(w, fma(w, fma(w, G7, G5), G3), G1);
  double R = odd + even;
  double hfsq = 0.5 * f * f;
  return dk * LN2_HI - ((hfsq - fma(s, hfsq + R, dk * LN2_LO)) - f);
}

static double kernel_sin(double x) {
  static const double S[6] = {-1.66666666666666324348e-01, 8.33333333332248946124e-03,
                              -1.98412698298579493134e-04, 2.75573137070700676789e-06,
                              -2.50507602534068634195e-08, 1.58969099521155010221e-10};
  double z = x * x;
  double p = S[5];
  for (int i = 4; i >= 0; --i) p = fma(z, p, S[i]);
  return fma(z * x, p, x);
}
static double kernel_cos(double x) {
  static const double C[6] = {4.16666666666666019037e-02, -1.38888888888741095749e-03,
                              2.48015872894767294178e-05, -2.75573143513906633035e-07,
                              2.08757232129817482790e-09, -1.13596475577881948265e-11};
  double z = x * x;
  double p = C[5];
  for (int i = 4; i >= 0; --i) p = fma(z, p, C[i]);
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * z) * p);
}

void orc_sincos(double x, double* s, double* c) {
  double fn = rint(x * 6.36619772367581382433e-01);
  int n = (int)fn;
  double r = x - fn * 1.57079632673412561417e+00;
  r = r - fn * 6.07710050630396597660e-11;
  r = r - fn * 2.02226624879595063154e-21;
  double ks = kernel_sin(r), kc = kernel_cos(r);
  switch (n & 3) {
    case 0: *s = ks; *c = kc; break;
    case 1: *s = kc; *c = -ks; break;
    case 2: *s = -ks; *c = -kc; break;
    default: *s = -kc; *c = ks; break;
  }
}

void orc_sincos_2pi_u32(uint32_t w, double* s, double* c) {
  uint32_t oct = w >> 29, frac = w & 0x1fffffffu;
  if (oct & 1u) frac = 0x20000000u - frac;
  double phi = ((double)frac * (1.0 / 536870912.0)) * 7.85398163397448279e-01;
  double ks = kernel_sin(phi), kc = kernel_cos(phi);
  switch (oct) {
    case 0: *s = ks; *c = kc; break;
    case 1: *s = kc; *c = ks; break;
    case 2: *s = kc; *c = -ks; break;
    case 3: *s = ks; *c = -kc; break;
    case 4: *s = -ks; *c = -kc; break;
    case 5: *s = -kc; *c = -ks; break;
    case 6: *s = -kc; *c = ks; break;
    default: *s = -ks; *c = kc; break;
  }
}

void orc_normal_pair(uint32_t w0, uint32_t w1, uint32_t w2, double* n0, double* n1) {
  uint64_t a = (((uint64_t)w1 << 32) | w0) >> 11;
  double u = (double)(a + 1) * (1.0 / 9007199254740992.0);
  double radius = sqrt(-2.0 * orc_log(u));
  double s, c;
  orc_sincos_2pi_u32(w2, &s, &c);
  *n0 = radius * c;
  *n1 = radius * s;
}

void orc_normal3(uint64_t seed, uint64_t index, uint32_t waypoint, uint32_t stream, double z[3],
                 uint32_t* spare) {
  uint32_t a[4], b[4];
  double drop;
  draw(seed, index, waypoint, stream, 0, a);
  draw(seed, index, waypoint, stream, 1, b);
  orc_normal_pair(a[0], a[1], a[2], &z[0], &z[1]);
  orc_normal_pair(b[0], b[1], b[2], &z[2], &drop);
  *spare = a[3];
}

/* ---- table-driven forms (hot path: mixture samples, footprint heading), numerics v9 ------------
 * lg[i] = {1/c_i rounded, 2 log(that) + 2 ln 2}, c_i = 1 + (i + 1/2)/512; sc[s] = {cos, sin} of 2 pi s / 256,
 * the sector boundaries.  Built from the functions above, on first use. */
static const double TWO_LN2 = 1.386294361119890572454e+00;
static double tab_lg[512][2], tab_sc[256][2];
static int tab_ready = 0;
static void tables(void) {
  if (tab_ready) return;
  for (int i = 0; i < 512; ++i) {
    double c = 1.0 + ((double)i + 0.5) / 512.0;
    tab_lg[i][0] = 1.0 / c;
    tab_lg[i][1] = 2.0 * orc_log(tab_lg[i][0]) + TWO_LN2;
  }
  for (int s = 0; s < 256; ++s) {
    double sn, cs;
    orc_sincos_2pi_u32((uint32_t)s << 24, &sn, &cs);
    tab_sc[s][0] = cs; tab_sc[s][1] = sn;
  }
  tab_ready = 1;
}

/* -2 log(w 2^-32), the squared Box-Muller radius of a uniform with the levels 0, 2^-32, ..., 1 - 2^-32:
 * w = 2^e t, e = 31 - (leading zeros z of w); cell i of t; r = t / c_i - 1; -2 log t = 2 log(1/c_i) - 2 log1p(r),
 * the latter as r times the cubic (-2 + r (1 + r (-2/3 + r/2))); the exponent's share is (z + 1) 2 ln 2, whose
 * "+ 1" sits in the table entry: z 2 ln 2 is added to the entry in one fma, the polynomial in a second.  The level
 * 0 has no logarithm: the build spec gives it the value of cell 0 with r = -1 and z = -1 (4.1647: a radius of 2.04). */
double orc_radius2_unit32(uint32_t w) {
  tables();
  double r, base;
  int z;
  if (w == 0) {
    r = -1.0; base = tab_lg[0][1]; z = -1;
  } else {
    double x = (double)w;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)(bits >> 52) - 1023;
    int i = (int)((bits >> 43) & 511);
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double t;
    memcpy(&t, &bits, 8);
    r = fma(t, tab_lg[i][0], -1.0);
    base = tab_lg[i][1];
    z = 31 - e;
  }
  static const double co[4] = {-2.0, 1.0, -2.0 / 3.0, 0.5};
  double p = co[3];
  for (int k = 2; k >= 0; --k) p = fma(r, p, co[k]);
  return fma(r, p, fma((double)z, TWO_LN2, base));
}

/* |d| <= pi/256 in radians: sine to d^5; cosine 1 - d^2/2 + C4 d^4 with the build spec's fitted C4 (the
 * coefficient of least maximum error on the interval, 5.0e-16; tools/make_v9_constants.py) */
static const double COS_C4 = 4.166647965169937434249e-02;
static void sincos_small(double d, double* sd, double* cd) {
  double z = d * d;
  double ps = fma(z, 1.0 / 120.0, -1.0 / 6.0);
  *sd = fma(d * z, ps, d);
  *cd = fma(z, fma(z, COS_C4, -0.5), 1.0);
}

/* the Box-Muller angle of a word: table entry of its top 8 bits, the low 24 bits READ AS A SIGNED (two's
 * complement) number k in [-2^23, 2^23) an offset from it -- a one-to-one map of words to the 2^32 angles --;
 * the same two polynomials in k instead of d = k a, a = 2 pi 2^-32, their coefficients scaled by powers of a */
void orc_sincos_2pi_u32_tab(uint32_t w, double* s, double* c) {
  static const double S1 = 1.462918079267159624024e-09, S3 = -5.218056424438286096208e-28,
                      S5 = 5.583657738838274755966e-47, C2 = -1.070064653323357779997e-18,
                      C4 = 1.908388704914255060734e-37;
  tables();
  int sec = (int)(w >> 24);
  int low = (int)(w & 0x00ffffffu);
  double k = (double)(low >= (1 << 23) ? low - (1 << 24) : low);
  double z = k * k;
  double sd = k * fma(z, fma(z, S5, S3), S1);
  double cd = fma(z, fma(z, C4, C2), 1.0);
  double C = tab_sc[sec][0], S = tab_sc[sec][1];
  *s = fma(S, cd, C * sd);
  *c = fma(C, cd, -(S * sd));
}

void orc_sincos_tab(double x, double* s, double* c) {
  tables();
  double fn = rint(x * 4.07436654315252084757e+01);       /* 256 / (2 pi) */
  int n = (int)fn;
  double d = fma(-fn, 2.45436926061702587187e-02, x);     /* pi/128 rounded: one step */
  double sd, cd;
  sincos_small(d, &sd, &cd);
  int sec = n & 255;
  double C = tab_sc[sec][0], S = tab_sc[sec][1];
  *s = fma(S, cd, C * sd);
  *c = fma(C, cd, -(S * sd));
}

/* Box-Muller pair of the mixture sampler: radius word wr, angle word wa. */
void orc_normal_pair_w2(uint32_t wr, uint32_t wa, double* n0, double* n1) {
  double radius = sqrt(fabs(orc_radius2_unit32(wr)));
  double s, c;
  orc_sincos_2pi_u32_tab(wa, &s, &c);
  *n0 = radius * c;
  *n1 = radius * s;
}

/* mixture samples come in pairs (2j, 2j+1) that share two draws keyed by the pair index j:
 *   slot 0: words (0,1) -> z0,z1 of 2j;  words (2,3) -> z2 of 2j, z0 of 2j+1
 *   slot 1: words (0,1) -> z1,z2 of 2j+1;  word 2 = spare of 2j, word 3 = spare of 2j+1.
 * This returns the normals and spare word of ONE sample. */
void orc_sample_normals(uint64_t seed, uint64_t sample, uint32_t waypoint, uint32_t stream,
                        double z[3], uint32_t* spare) {
  uint64_t pair = sample >> 1;
  uint32_t w0[4], w1[4];
  double a, b;
  draw(seed, pair, waypoint, stream, 0, w0);
  draw(seed, pair, waypoint, stream, 1, w1);
  orc_normal_pair_w2(w0[2], w0[3], &a, &b);
  if ((sample & 1) == 0) {
    orc_normal_pair_w2(w0[0], w0[1], &z[0], &z[1]);
    z[2] = a;
    *spare = w1[2];
  } else {
    z[0] = b;
    orc_normal_pair_w2(w1[0], w1[1], &z[1], &z[2]);
    *spare = w1[3];
  }
}

/* ---- component counts: Multinomial(N, weights) by conditional binomials ------------------------
 * The reference counts N categorical draws (GM_Model.h:87-93).  Bin(n, p): waiting times when
 * n min(p,1-p) < 10, else BTPE (Kachitvichyanukul & Schmeiser, CACM 31(2), 1988).  Uniforms: 53 bits
 * of two words of a draw on stream 4, counter = (component, 0, waypoint, 4 << 16 | draw number). */
typedef struct { uint64_t seed; uint32_t comp, waypoint, draw; uint32_t w[4]; int have; } count_rng;

static double count_uniform(count_rng* g) {
  if (g->have == 0) { draw(g->seed, (uint64_t)g->comp, g->waypoint, 4, g->draw, g->w); g->draw += 1; g->have = 2; }
  uint32_t lo = (g->have == 2) ? g->w[0] : g->w[2], hi = (g->have == 2) ? g->w[1] : g->w[3];
  g->have -= 1;
  uint64_t m = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)m + 0.5) * (1.0 / 9007199254740992.0);
}

double orc_binomial(double n, double p, uint64_t seed, uint32_t comp, uint32_t waypoint) {
  count_rng g; g.seed = seed; g.comp = comp; g.waypoint = waypoint; g.draw = 0; g.have = 0;
  if (!(n >= 1.0) || !(p > 0.0)) return 0.0;
  if (p >= 1.0) return n;
  int flip = p > 0.5;
  double r = flip ? 1.0 - p : p, q = 1.0 - r, y;
  if (n * r < 10.0) {
    double lq = orc_log(q), pos = 0.0;
    y = 0.0;
    for (int it = 0; it < 400; ++it) {
      pos += floor(orc_log(count_uniform(&g)) / lq) + 1.0;
      if (pos > n) break;
      y += 1.0;
    }
    return flip ? n - y : y;
  }
  double nrq = n * r * q, fm = n * r + r, m = floor(fm);
  double p1 = floor(2.195 * sqrt(nrq) - 4.6 * q) + 0.5;
  double xm = m + 0.5, xl = xm - p1, xr = xm + p1;
  double c = 0.134 + 20.5 / (15.3 + m);
  double a = (fm - xl) / (fm - xl * r);
  double laml = a * (1.0 + a / 2.0);
  a = (xr - fm) / (xr * q);
  double lamr = a * (1.0 + a / 2.0);
  double p2 = p1 * (1.0 + 2.0 * c), p3 = p2 + c / laml, p4 = p3 + c / lamr;
  y = m;
  for (int it = 0; it < 1000; ++it) {
    double u = count_uniform(&g) * p4;
    double v = count_uniform(&g);
    if (u <= p1) { y = floor(xm - p1 * v + u); break; }
    if (u <= p2) {
      double x = xl + (u - p1) / c;
      v = v * c + 1.0 - fabs(m - x + 0.5) / p1;
      if (v > 1.0) continue;
      y = floor(x);
    } else if (u <= p3) {
      y = floor(xl + orc_log(v) / laml);
      if (y < 0.0) continue;
      v = v * (u - p2) * laml;
    } else {
      y = floor(xr - orc_log(v) / lamr);
      if (y > n) continue;
      v = v * (u - p3) * lamr;
    }
    double k = fabs(y - m);
    if (k > 20.0 && k < nrq / 2.0 - 1.0) {
      double rho = (k / nrq) * ((k * (k / 3.0 + 0.625) + 0.16666666666666666) / nrq + 0.5);
      double t = -k * k / (2.0 * nrq);
      double A = orc_log(v);
      if (A < t - rho) break;
      if (A > t + rho) continue;
      double x1 = y + 1.0, f1 = m + 1.0, z = n + 1.0 - m, w = n - y + 1.0;
      double x2 = x1 * x1, f2 = f1 * f1, z2 = z * z, w2 = w * w;
      double bound = xm * orc_log(f1 / x1) + (n - m + 0.5) * orc_log(z / w) + (y - m) * orc_log(w * r / (x1 * q)) +
                     (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / f2) / f2) / f2) / f2) / f1 / 166320.0 +
                     (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / z2) / z2) / z2) / z2) / z / 166320.0 +
                     (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / x2) / x2) / x2) / x2) / x1 / 166320.0 +
                     (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / w2) / w2) / w2) / w2) / w / 166320.0;
      if (A > bound) continue;
      break;
    }
    double s = r / q, aa = s * (n + 1.0), F = 1.0;
    if (m < y) { for (double i = m + 1.0; i <= y; i += 1.0) F *= (aa / i - s); }
    else if (m > y) { for (double i = y + 1.0; i <= m; i += 1.0) F /= (aa / i - s); }
    if (v > F) continue;
    break;
  }
  return flip ? n - y : y;
}

/* ------------------------------------------------------------------------------------------ */
/* estimator math                                                                              */
/* ------------------------------------------------------------------------------------------ */
#define ORC_PI 3.14159265358979323846 /* the reference's MPI macro, MCSimulator.h:43 */

double orc_wrap_angle(double angle) { /* MCSimulator.h:56-65 */
  if (!(fabs(angle) <= 1.0e9)) return angle; /* guard: build spec */
  while (angle < 0) angle += 2 * ORC_PI;
  while (angle > (2 * ORC_PI)) angle -= 2 * ORC_PI;
  return angle;
}

void orc_prediction(const double state[3], const double cmd[3], double out[3]) { /* :413-431 */
  double drot1 = cmd[0], dtrans = cmd[1], drot2 = cmd[2];
  double s, c;
  orc_sincos(state[2] + drot1, &s, &c);
  out[0] = fma(dtrans, c, state[0]);
  out[1] = fma(dtrans, s, state[1]);
  out[2] = orc_wrap_angle(state[2] + drot1 + drot2);
}

void orc_inverse_odometry(const double p1[3], const double p2[3], double out[3]) { /* :434-449 */
  double drot1 = atan2(p2[1] - p1[1], p2[0] - p1[0]) - p1[2];
  drot1 = orc_wrap_angle(drot1);
  double ex = p2[0] - p1[0], ey = p2[1] - p1[1];
  double dtrans = sqrt(ex * ex + ey * ey);
  double drot2 = p2[2] - p1[2] - drot1;
  drot2 = orc_wrap_angle(drot2);
  out[0] = drot1; out[1] = dtrans; out[2] = drot2;
}

void orc_generate_M(const double alphas[4], const double cmd[3], double Mdiag[3]) { /* :495-513 */
  double r1 = cmd[0], tr = cmd[1], r2 = cmd[2];
  Mdiag[0] = alphas[0] * (r1 * r1) + alphas[1] * (tr * tr);
  Mdiag[1] = alphas[2] * (tr * tr) + alphas[3] * (r1 * r1) + alphas[3] * (r2 * r2);
  Mdiag[2] = alphas[0] * (r2 * r2) + alphas[1] * (tr * tr);
}

typedef double mat3[3][3];

static void mm(mat3 A, mat3 B, mat3 C) { /* C = A B, inner sum left to right */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = (A[i][0] * B[0][j] + A[i][1] * B[1][j]) + A[i][2] * B[2][j];
}
static void mmt(mat3 A, mat3 B, mat3 C) { /* C = A B^T */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = (A[i][0] * B[j][0] + A[i][1] * B[j][1]) + A[i][2] * B[j][2];
}

/* EKFpredict :868-881.  Sigma, predSigma row-major 9. */
void orc_ekf_predict(const double mu[3], const double Sigma[9], const double u[3],
                     const double Mdiag[3], double predMu[3], double predSigma[9]) {
  double s, c;
  orc_sincos(mu[2] + u[0], &s, &c);
  mat3 G = {{1, 0, -u[1] * s}, {0, 1, u[1] * c}, {0, 0, 1}};            /* :517-529 */
  mat3 V = {{-u[1] * s, c, 0}, {u[1] * c, s, 0}, {1, 0, 1}};            /* :453-468, as written */
  mat3 M = {{Mdiag[0], 0, 0}, {0, Mdiag[1], 0}, {0, 0, Mdiag[2]}};
  mat3 S, VM, R, GS, GSG;
  memcpy(S, Sigma, sizeof S);
  mm(V, M, VM);
  mmt(VM, V, R);                                                         /* R = V M V^T */
  mm(G, S, GS);
  mmt(GS, G, GSG);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) predSigma[3 * i + j] = GSG[i][j] + R[i][j];
  predMu[0] = fma(u[1], c, mu[0]);                                       /* prediction(mu,u) */
  predMu[1] = fma(u[1], s, mu[1]);
  predMu[2] = orc_wrap_angle(mu[2] + u[0] + u[2]);
}

/* EKFupdate :883-929, in place on (mu, Sigma); landmarks lx,ly; measurements z[L]. */
void orc_ekf_update(double mu[3], double Sigma[9], const double* z, int L, const double* lx,
                    const double* ly, double Q) {
  for (int lid = 0; lid < L; ++lid) {
    double diff0 = mu[0] - lx[lid], diff1 = mu[1] - ly[lid];           /* makeHRow :470-492 */
    double q = diff0 * diff0 + diff1 * diff1;
    double root = sqrt(q);
    double H[3] = {-(lx[lid] - mu[0]) / root, -(ly[lid] - mu[1]) / root, 0.0};
    mat3 P;
    memcpy(P, Sigma, sizeof P);
    double HP0 = H[0] * P[0][0] + H[1] * P[1][0];                      /* H * Sigma (H[2] = 0) */
    double HP1 = H[0] * P[0][1] + H[1] * P[1][1];
    double S = (HP0 * H[0] + HP1 * H[1]) + Q;                           /* :902 */
    double Sinv = 1.0 / S;                                              /* 1x1 .i() */
    double K[3];
    for (int i = 0; i < 3; ++i) K[i] = (P[i][0] * H[0] + P[i][1] * H[1]) * Sinv;   /* :906 */
    double zhat = root;                                                 /* observation :368-381 */
    double innovation = z[lid] - zhat;
    for (int i = 0; i < 3; ++i) mu[i] = mu[i] + K[i] * innovation;      /* :919 */
    mat3 A, N;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A[i][j] = (i == j ? 1.0 : 0.0) - K[i] * H[j];
    A[0][2] = 0.0; A[1][2] = 0.0; A[2][2] = 1.0;                        /* H[2] == 0 exactly */
    mm(A, P, N);                                                        /* :921 */
    memcpy(Sigma, N, sizeof N);
  }
}

/* chol(C,"lower"), lower triangle only (LAPACK potrf semantics). L = l00 l10 l11 l20 l21 l22 */
int orc_chol3_lower(const double C[9], double L[6]) {
  if (!(C[0] > 0.0)) return 0;
  double l00 = sqrt(C[0]);
  double l10 = C[3] / l00, l20 = C[6] / l00;
  double t = C[4] - l10 * l10;
  if (!(t > 0.0)) return 0;
  double l11 = sqrt(t);
  double l21 = (C[7] - l20 * l10) / l11;
  double v = (C[8] - l20 * l20) - l21 * l21;
  if (!(v > 0.0)) return 0;
  L[0] = l00; L[1] = l10; L[2] = l11; L[3] = l20; L[4] = l21; L[5] = sqrt(v);
  return 1;
}

/* mvnrnd(M, C, n) on a GIVEN tape of standard normals (glue_mvnrnd_meat.hpp:134-145: out = D * randn(3, n),
 * each column += M, D = chol(C, "lower")).  z / out: 3 x n column-major.  The same three fma chains
 * orc_gmm_waypoint applies to its own draws; exists so that the transform can be compared with the
 * reference's mvnrnd run on the same tape (tests/test_oracle_vs_ref_lapack.py).  Returns 0 when the
 * factorisation fails (the reference then takes its eigen-decomposition path, :100-132). */
int orc_mvnrnd_tape(const double mean[3], const double C[9], const double* z, long long n, double* out) {
  double L[6];
  if (!orc_chol3_lower(C, L)) return 0;
  for (long long i = 0; i < n; ++i) {
    const double* zz = z + 3 * i;
    double* pt = out + 3 * i;
    pt[0] = fma(L[0], zz[0], mean[0]);
    pt[1] = fma(L[2], zz[1], fma(L[1], zz[0], mean[1]));
    pt[2] = fma(L[5], zz[2], fma(L[4], zz[1], fma(L[3], zz[0], mean[2])));
  }
  return 1;
}

/* arma::mean(X,1) and arma::cov(X.t()) for X = 3 x n given as n rows of 3 (op_mean_meat.hpp:104-133,
 * op_cov_meat.hpp:26-53): acc = sum(A); out = A^T A; out -= acc^T acc / N; out /= (N-1). */
void orc_cov_mean(const double* rows, long long n, double mean[3], double cov[9]) {
  double acc[3] = {0, 0, 0};
  double ata[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (long long r = 0; r < n; ++r) {
    const double* p = rows + 3 * r;
    for (int i = 0; i < 3; ++i) {
      acc[i] += p[i];
      for (int j = 0; j < 3; ++j) ata[i][j] += p[i] * p[j];
    }
  }
  double N = (double)n;
  double norm_val = (n > 1) ? (N - 1.0) : 1.0;
  for (int i = 0; i < 3; ++i) {
    mean[i] = acc[i] / N;
    for (int j = 0; j < 3; ++j) cov[3 * i + j] = (ata[i][j] - (acc[i] * acc[j]) / N) / norm_val;
  }
}

/* normalise(row, 1): divide by the L1 norm; zero norm divides by 1 (op_normalise_meat.hpp:107-121) */
void orc_normalise_l1(const double* in, int n, double* out) {
  double norm = 0.0;
  for (int i = 0; i < n; ++i) norm += fabs(in[i]);
  double d = (norm != 0.0) ? norm : 1.0;
  for (int i = 0; i < n; ++i) out[i] = in[i] / d;
}

/* ------------------------------------------------------------------------------------------ */
/* collision predicate (build spec): footprint box vs static boxes, separating axes            */
/* fp = {dx, dy, hx, hy}; boxes = M x {cx, cy, hx, hy, yaw}                                      */
/* ------------------------------------------------------------------------------------------ */
int orc_collides(double x, double y, double th, const double fp[4], const double* boxes, int M) {
  double s, c;
  orc_sincos_tab(th, &s, &c);
  double px = x + fma(c, fp[0], -(s * fp[1]));
  double py = y + fma(s, fp[0], c * fp[1]);
  double rx = fp[2], ry = fp[3];
  double rr = sqrt(rx * rx + ry * ry);
  int hit = 0;
  for (int m = 0; m < M; ++m) {
    const double* b = boxes + 5 * m;
    double bs, bc;
    orc_sincos(b[4], &bs, &bc);
    double hx = b[2], hy = b[3];
    double reach_x = (hx * fabs(bc) + hy * fabs(bs)) + rr;   /* inflated world AABB */
    double reach_y = (hx * fabs(bs) + hy * fabs(bc)) + rr;
    double ddx = b[0] - px, ddy = b[1] - py;
    if (fabs(ddx) > reach_x || fabs(ddy) > reach_y) continue;
    double cr = fabs(fma(c, bc, s * bs));
    double sr = fabs(fma(s, bc, -(c * bs)));
    double in_robot_1 = fma(ddx, c, ddy * s), in_robot_2 = fma(ddy, c, -(ddx * s));
    double in_box_1 = fma(ddx, bc, ddy * bs), in_box_2 = fma(ddy, bc, -(ddx * bs));
    if (fabs(in_robot_1) > rx + fma(hx, cr, hy * sr)) continue;
    if (fabs(in_robot_2) > ry + fma(hx, sr, hy * cr)) continue;
    if (fabs(in_box_1) > hx + fma(rx, cr, ry * sr)) continue;
    if (fabs(in_box_2) > hy + fma(rx, sr, ry * cr)) continue;
    hit = 1;
  }
  return hit;
}

/* ------------------------------------------------------------------------------------------ */
/* configuration shared by the run functions                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  double alphas[4];
  double Q;
  int L;
  int W;
  int K;
  int M;
  double lx[ORC_MAX_L], ly[ORC_MAX_L];
  double cov0[9];
  double fp[4];
  const double* traj;   /* 3 x W by component  (setTrajectory layout, mcsimplugin.cpp:83-97)  */
  const double* odom;   /* 3 x (W-1) by component (mcsimplugin.cpp:99-113)                    */
  const double* boxes;  /* M x 5 */
} orc_config;

/* Test hook (tests/test_oracle_vs_ref_loop.py): standard normals from TAPES instead of Philox, so that a run can be
 * repeated on the noise the reference's compiled loop (oracle/_ref/libpocs_ref_loop.so) drew from arma::randn and
 * compared value for value.  chain: (W-1) x stride, a step's r1 tr r2 z_0..z_{L-1}; init: 3 per MC particle; gmm:
 * per waypoint 3 per sample, n_gmm samples a waypoint; counts: per waypoint K component counts (the reference draws
 * them from GM_Model's own engine and prints them), replacing the conditional binomials.  NULL = that stream stays
 * on Philox.  Not thread-safe. */
static const double *g_tape_chain = NULL, *g_tape_init = NULL, *g_tape_gmm = NULL;
static const long long* g_tape_counts = NULL;
static int g_tape_stride = 0;
static long long g_tape_n_gmm = 0;
void orc_set_tapes(const double* chain, int stride, const double* init, const double* gmm, long long n_gmm,
                   const long long* counts) {
  g_tape_chain = chain; g_tape_stride = stride; g_tape_init = init; g_tape_gmm = gmm; g_tape_n_gmm = n_gmm;
  g_tape_counts = counts;
}

/* orc_collides on a configuration's footprint and boxes, in the shape of a callback (the compiled reference loop of
 * oracle/ref_loop_harness.cpp takes its collision predicate this way). */
int orc_collides_cfg(double x, double y, double th, const void* cfg_) {
  const orc_config* cfg = (const orc_config*)cfg_;
  return orc_collides(x, y, th, cfg->fp, cfg->boxes, cfg->M);
}

static double chain_normal(uint64_t seed, int step, int k) {
  if (g_tape_chain) return g_tape_chain[(size_t)step * (size_t)g_tape_stride + (size_t)k];
  uint32_t w[4];
  double a, b;
  draw(seed, (uint64_t)step, 0, 1 /* chain stream */, (uint32_t)(k / 2), w);
  orc_normal_pair(w[0], w[1], w[2], &a, &b);
  return (k % 2) ? b : a;
}

/* EKF_GaussProp's particle-independent part, MCSimulator.h:692-800.  Per step i < W-1 writes
 * applied[3i..], Mdiag[3i..], noisy[3i..], z[L*i..], mu[3i..], cov[9i..] (any may be NULL). */
/* generateL (MCSimulator.h:532-553) and the applied control EKF_GaussProp forms from it (:714-726):
 * gain = diag(ubar_j / (xhat_j != 0 ? xhat_j : 0.1)), ubar = inverseOdometry(estimated, goal) - u*,
 * xhat = estimated - nominal; applied = u* + gain xhat (the off-diagonal zeros of the 3 x 3 product add
 * exact zeros).  Pinned against the reference's own text: tests/test_oracle_vs_ref_ekf.py. */
void orc_applied_control(const double nominal[3], const double estimated[3], const double goal[3],
                         const double control[3], double gain[3], double applied[3]) {
  double urequired[3];
  orc_inverse_odometry(estimated, goal, urequired);                      /* :537 */
  for (int j = 0; j < 3; ++j) {
    double xhat = estimated[j] - nominal[j];
    double ubar = urequired[j] - control[j];
    gain[j] = ubar / (xhat != 0 ? xhat : 0.1);                           /* :548-550 */
    applied[j] = control[j] + gain[j] * xhat;                            /* :722-726 */
  }
}

int orc_host_chain(const orc_config* cfg, uint64_t seed, double* applied, double* Mdiag,
                   double* noisy, double* z, double* mu_out, double* cov_out) {
  int W = cfg->W, L = cfg->L;
  double mu[3] = {cfg->traj[0], cfg->traj[W], cfg->traj[2 * W]};     /* initialmu = col 0, :164 */
  double cov[9];
  memcpy(cov, cfg->cov0, sizeof cov);
  double realstate[3] = {mu[0], mu[1], mu[2]};
  for (int i = 0; i < W - 1; ++i) {
    double control[3], nominal[3], goal[3];
    for (int j = 0; j < 3; ++j) {
      control[j] = cfg->odom[j * (W - 1) + i];
      nominal[j] = cfg->traj[j * W + i];
      goal[j] = cfg->traj[j * W + i + 1];
    }
    double Md[3];
    orc_generate_M(cfg->alphas, control, Md);                            /* :701 nominal control */
    double gain[3], appliedc[3];
    orc_applied_control(nominal, mu, goal, control, gain, appliedc);     /* generateL :532-553, :714-726 */
    double predMu[3], predSigma[9];
    orc_ekf_predict(mu, cov, appliedc, Md, predMu, predSigma);           /* :746 */
    double a1 = cfg->alphas[0], a2 = cfg->alphas[1], a3 = cfg->alphas[2], a4 = cfg->alphas[3];
    double r1 = appliedc[0], tr = appliedc[1], r2 = appliedc[2];
    double var0 = a1 * (r1 * r1) + a2 * (tr * tr);                       /* sampleOdometry :403-405 */
    double var1 = a3 * (tr * tr) + a4 * ((r1 * r1) + (r2 * r2));
    double var2 = a1 * (r2 * r2) + a2 * (tr * tr);
    double noisyc[3];
    noisyc[0] = r1 + chain_normal(seed, i, 0) * sqrt(var0);              /* sampleNormal :51-53 */
    noisyc[1] = tr + chain_normal(seed, i, 1) * sqrt(var1);
    noisyc[2] = r2 + chain_normal(seed, i, 2) * sqrt(var2);
    double nextstate[3];
    orc_prediction(realstate, noisyc, nextstate);                        /* :407 */
    memcpy(realstate, nextstate, sizeof realstate);
    double obs[ORC_MAX_L];
    for (int l = 0; l < L; ++l) {                                        /* :786-789 */
      double ex = realstate[0] - cfg->lx[l], ey = realstate[1] - cfg->ly[l];
      double distance = sqrt(ex * ex + ey * ey);
      obs[l] = distance + (0.0 + chain_normal(seed, i, 3 + l) * sqrt(cfg->Q));
    }
    orc_ekf_update(predMu, predSigma, obs, L, cfg->lx, cfg->ly, cfg->Q);  /* :797 */
    memcpy(mu, predMu, sizeof mu);
    memcpy(cov, predSigma, sizeof cov);
    if (applied) memcpy(applied + 3 * i, appliedc, sizeof appliedc);
    if (Mdiag) memcpy(Mdiag + 3 * i, Md, sizeof Md);
    if (noisy) memcpy(noisy + 3 * i, noisyc, sizeof noisyc);
    if (z) memcpy(z + (size_t)L * i, obs, (size_t)L * sizeof(double));
    if (mu_out) memcpy(mu_out + 3 * i, mu, sizeof mu);
    if (cov_out) memcpy(cov_out + 9 * i, cov, sizeof cov);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* MC path: runSimulation, MCSimulator.h:361-365 -> EKF_GaussProp("MC")                          */
/* Particles [first, first+count) of the N configured.  hits (count) and final particles       */
/* (count x 3, x y theta triples = the reference's 3 x N column-major) may be NULL.            */
/* Returns the number of particles with hits > 0 (getCollisionProportion's numerator).         */
/* ------------------------------------------------------------------------------------------ */
long long orc_run_mc(const orc_config* cfg, uint64_t seed, long long first, long long count,
                     uint32_t* hits_out, double* particles_out) {
  int W = cfg->W;
  double* noisy = (double*)malloc(sizeof(double) * 3 * (size_t)(W > 1 ? W - 1 : 1));
  orc_host_chain(cfg, seed, NULL, NULL, noisy, NULL, NULL, NULL);
  double L0[6];
  if (!orc_chol3_lower(cfg->cov0, L0)) { free(noisy); return -1; }
  double mu0[3] = {cfg->traj[0], cfg->traj[W], cfg->traj[2 * W]};
  long long collided = 0;
  for (long long i = 0; i < count; ++i) {
    double zz[3];
    uint32_t spare;
    orc_normal3(seed, (uint64_t)(first + i), 0, 2 /* mc-init stream */, zz, &spare);
    if (g_tape_init) memcpy(zz, g_tape_init + 3 * (size_t)(first + i), sizeof zz);
    double p[3];                                                         /* initParticles :287-297 */
    p[0] = fma(L0[0], zz[0], mu0[0]);
    p[1] = fma(L0[2], zz[1], fma(L0[1], zz[0], mu0[1]));
    p[2] = fma(L0[5], zz[2], fma(L0[4], zz[1], fma(L0[3], zz[0], mu0[2])));
    uint32_t h = orc_collides(p[0], p[1], p[2], cfg->fp, cfg->boxes, cfg->M) ? 1u : 0u;   /* :668 */
    for (int s = 0; s < W - 1; ++s) {                                    /* :760-761 */
      double q[3];
      orc_prediction(p, noisy + 3 * s, q);                               /* moveParticles :300-322 */
      memcpy(p, q, sizeof p);
      if (orc_collides(p[0], p[1], p[2], cfg->fp, cfg->boxes, cfg->M)) ++h;
    }
    if (h > 0) ++collided;
    if (hits_out) hits_out[i] = h;
    if (particles_out) memcpy(particles_out + 3 * i, p, sizeof p);
  }
  free(noisy);
  return collided;
}

/* ------------------------------------------------------------------------------------------ */
/* GMM path                                                                                    */
/* state (per component, ORC_STATE doubles): mean[3] cov[9] weight alive pad pad               */
/* moments (per component, ORC_NMOM): nFree nColl Sx Sy St Sxx Sxy Sxt Syy Syt Stt             */
/* ------------------------------------------------------------------------------------------ */

/* One waypoint's sampleNPoints + checkMatrixCollisions + sums for samples [first, first+count):
 * GM_Model.h:83-116 and MCSimulator.h:582-611.  The component of a sample is one categorical
 * draw per sample (GM_Model.h:89-93 draws N of them and counts; drawing it next to the sample
 * gives the same joint law and keeps a sample's randomness a function of its index).
 * samples_out (count x 3), flags_out (count), comp_out (count) optional. */
/* counts[k] of the n_total samples of a waypoint, as their running sum: Multinomial(n_total,
 * weights) like the reference's N categorical draws (GM_Model.h:87-93), drawn as conditional
 * binomials in component order; retired components get none; all retired: component 0 gets all. */
void orc_component_counts(int K, const double* state, uint64_t seed, int waypoint, long long n_total,
                          double* cumulative) {
  int last_alive = -1;
  double suffix[ORC_MAX_K], tail = 0.0;
  for (int k = K - 1; k >= 0; --k) {
    const double* st = state + k * ORC_STATE;
    int live = st[13] != 0.0 && st[12] > 0.0;
    if (live) { tail += st[12]; if (last_alive < 0) last_alive = k; }
    suffix[k] = tail;
  }
  double remaining = (double)n_total, running = 0.0;
  for (int k = 0; k < K; ++k) {
    const double* st = state + k * ORC_STATE;
    double nk = 0.0;
    if (last_alive < 0) nk = (k == 0) ? remaining : 0.0;
    else if (k == last_alive) nk = remaining;
    else if (k < last_alive && st[13] != 0.0 && st[12] > 0.0)
      nk = orc_binomial(remaining, st[12] / suffix[k], seed, (uint32_t)k, (uint32_t)waypoint);
    remaining -= nk;
    running += nk;
    cumulative[k] = running;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* The sums of truncateGMM (MCSimulator.h:592-611) over the free samples of each component.  The reference
 * takes mean(free,1) and cov(free^T) with Armadillo (op_mean / op_cov -> BLAS), whose summation order is
 * whatever the library's kernels do -- it has no canonical order.  The build fixes one (DESIGN.md section
 * 4, "summation tree", numerics v7), a function of the shard's sample count only:
 *   the shard's pairs of samples (2 lp, 2 lp + 1) in CHUNKS of 512 pairs; the chunks cut into
 *   VS = the largest power of two <= min(256, chunks) VIRTUAL SLICES, slice j = chunks
 *   [j chunks / VS, (j + 1) chunks / VS); pair lp = 512 c + 64 v + l sits in chunk c, WAVE v, LANE l;
 *   lane chain  the free samples of one component that lane l of wave v meets in slice j, in chunk
 *               order, sample 2 lp before 2 lp + 1: s += x ..., sxx = fma(x, x, sxx) ...
 *   wave sum    the 64 lane chains: g_q = lanes 8q .. 8q+7 added in lane order, then
 *               ((g0 + g1) + (g2 + g3)) + ((g4 + g5) + (g6 + g7))
 *   row         of (slice, component): the eight wave sums in wave order
 *   total       sixteen interleaved partial sums P_g = row g + row (g + 16) + ..., then P_0 + ... + P_15
 * order 0 = the plain sequential sums over the free set in sample order (what this file did before v7;
 * kept to show how far two orders lie apart: tests/test_oracle_sum_orders.py).                          */
/* ------------------------------------------------------------------------------------------ */
static int g_sum_order = 1;
void orc_set_sum_order(int order) { g_sum_order = order ? 1 : 0; }
int orc_get_sum_order(void) { return g_sum_order; }

static void tree_moments(int K, long long count, const double* pts, const unsigned char* hit,
                         const unsigned char* comp, double* moments) {
  enum { TB = 512, LANES = 64, WAVES = TB / LANES, NS = 9 };
  long long npairs = (count + 1) / 2;
  long long chunks = (npairs + TB - 1) / TB;
  if (chunks < 1) chunks = 1;
  int sh = 0;
  while ((2LL << sh) <= chunks && (2 << sh) <= 256) ++sh;
  int VS = 1 << sh;
  double* rows = (double*)calloc((size_t)VS * ORC_MAX_K * (NS + 1), sizeof(double));   /* [slice][k][n, 9 sums] */
  double (*chain)[LANES][NS] = malloc(sizeof(double[ORC_MAX_K][LANES][NS]));           /* (not static: bench.py times the oracle on many threads) */
  for (int j = 0; j < VS; ++j) {
    long long cb = ((long long)j * chunks) >> sh, ce = ((long long)(j + 1) * chunks) >> sh;
    for (int v = 0; v < WAVES; ++v) {
      long long n_k[ORC_MAX_K];
      for (int k = 0; k < K; ++k) { n_k[k] = 0; memset(chain[k], 0, sizeof(double[LANES][NS])); }
      for (long long c = cb; c < ce; ++c)
        for (int l = 0; l < LANES; ++l) {
          long long lp = c * TB + (long long)v * LANES + l;
          for (int h = 0; h < 2; ++h) {
            long long i = 2 * lp + h;
            if (i >= count || hit[i]) continue;
            const double* p = pts + 3 * i;
            double* a = chain[comp[i]][l];
            ++n_k[comp[i]];
            a[0] += p[0]; a[1] += p[1]; a[2] += p[2];
            a[3] = fma(p[0], p[0], a[3]); a[4] = fma(p[0], p[1], a[4]); a[5] = fma(p[0], p[2], a[5]);
            a[6] = fma(p[1], p[1], a[6]); a[7] = fma(p[1], p[2], a[7]); a[8] = fma(p[2], p[2], a[8]);
          }
        }
      for (int k = 0; k < K; ++k) {
        double* row = rows + ((size_t)j * ORC_MAX_K + k) * (NS + 1);
        double wn = (double)n_k[k];
        row[0] = (v == 0) ? wn : row[0] + wn;
        for (int s = 0; s < NS; ++s) {
          double g[8];
          for (int q = 0; q < 8; ++q) {
            double t = chain[k][8 * q][s];
            for (int l = 1; l < 8; ++l) t += chain[k][8 * q + l][s];
            g[q] = t;
          }
          double ws = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
          row[1 + s] = (v == 0) ? ws : row[1 + s] + ws;
        }
      }
    }
  }
  for (int k = 0; k < K; ++k) {
    double* m = moments + k * ORC_NMOM;
    for (int s = 0; s <= NS; ++s) {
      double part[16];
      int ng = VS < 16 ? VS : 16;
      for (int g = 0; g < ng; ++g) {
        double t = 0.0;
        for (int q = g; q < VS; q += 16) t += rows[((size_t)q * ORC_MAX_K + k) * (NS + 1) + s];
        part[g] = t;
      }
      double tot = part[0];
      for (int g = 1; g < ng; ++g) tot += part[g];
      m[s == 0 ? 0 : s + 1] = tot;
    }
  }
  free(rows);
  free(chain);
}

int orc_gmm_waypoint(const orc_config* cfg, uint64_t seed, int waypoint, const double* state,
                     long long first, long long count, long long n_total, double* moments,
                     double* samples_out, int16_t* flags_out, int8_t* comp_out) {
  int K = cfg->K;
  double chol[ORC_MAX_K][6];
  double table[ORC_MAX_K];
  for (int k = 0; k < K; ++k) {
    const double* st = state + k * ORC_STATE;
    memset(chol[k], 0, sizeof chol[k]);
    if (st[13] != 0.0) orc_chol3_lower(st + 3, chol[k]);
  }
  orc_component_counts(K, state, seed, waypoint, n_total, table);
  if (g_tape_counts) {
    double run = 0.0;
    for (int k = 0; k < K; ++k) { run += (double)g_tape_counts[(size_t)waypoint * (size_t)K + (size_t)k]; table[k] = run; }
  }
  /* the reference keeps one matrix of points per component (GM_Model.h:99-107) and takes
   * mean / cov of the free columns afterwards (MCSimulator.h:592-598); do the same. */
  double* free_pts[ORC_MAX_K];
  long long nfree[ORC_MAX_K], ncoll[ORC_MAX_K];
  for (int k = 0; k < K; ++k) {
    free_pts[k] = (double*)malloc(sizeof(double) * 3 * (size_t)(count > 0 ? count : 1));
    nfree[k] = 0; ncoll[k] = 0;
  }
  double* all_pts = (double*)malloc(sizeof(double) * 3 * (size_t)(count > 0 ? count : 1));
  unsigned char* all_hit = (unsigned char*)malloc((size_t)(count > 0 ? count : 1));
  unsigned char* all_k = (unsigned char*)malloc((size_t)(count > 0 ? count : 1));
  for (long long i = 0; i < count; ++i) {
    double zz[3];
    uint32_t spare;
    orc_sample_normals(seed, (uint64_t)(first + i), (uint32_t)waypoint, 3 /* gmm stream */, zz, &spare);
    (void)spare;
    if (g_tape_gmm) memcpy(zz, g_tape_gmm + 3 * ((size_t)waypoint * (size_t)g_tape_n_gmm + (size_t)(first + i)), sizeof zz);
    double gidx = (double)(first + i);                /* component = first one whose running count exceeds the index */
    int k = 0;
    for (int j = 0; j < K - 1; ++j) if (table[j] <= gidx) ++k;
    const double* st = state + k * ORC_STATE;
    const double* Lk = chol[k];
    double pt[3];                                                       /* mvnrnd: D*z + M */
    pt[0] = fma(Lk[0], zz[0], st[0]);
    pt[1] = fma(Lk[2], zz[1], fma(Lk[1], zz[0], st[1]));
    pt[2] = fma(Lk[5], zz[2], fma(Lk[4], zz[1], fma(Lk[3], zz[0], st[2])));
    int hit = orc_collides(pt[0], pt[1], pt[2], cfg->fp, cfg->boxes, cfg->M);
    if (hit) ++ncoll[k];
    else { memcpy(free_pts[k] + 3 * nfree[k], pt, sizeof pt); ++nfree[k]; }
    memcpy(all_pts + 3 * i, pt, sizeof pt); all_hit[i] = (unsigned char)hit; all_k[i] = (unsigned char)k;
    if (samples_out) memcpy(samples_out + 3 * i, pt, sizeof pt);
    if (flags_out) flags_out[i] = (int16_t)hit;
    if (comp_out) comp_out[i] = (int8_t)k;
  }
  if (g_sum_order == 1) {
    tree_moments(K, count, all_pts, all_hit, all_k, moments);
    for (int k = 0; k < K; ++k) {
      moments[k * ORC_NMOM + 1] = (double)ncoll[k];
      free(free_pts[k]);
    }
    free(all_pts); free(all_hit); free(all_k);
    return 0;
  }
  for (int k = 0; k < K; ++k) {
    double* m = moments + k * ORC_NMOM;
    double sx = 0, sy = 0, st = 0, sxx = 0, sxy = 0, sxt = 0, syy = 0, syt = 0, stt = 0;
    for (long long r = 0; r < nfree[k]; ++r) {
      const double* p = free_pts[k] + 3 * r;
      sx += p[0]; sy += p[1]; st += p[2];
      sxx += p[0] * p[0]; sxy += p[0] * p[1]; sxt += p[0] * p[2];
      syy += p[1] * p[1]; syt += p[1] * p[2]; stt += p[2] * p[2];
    }
    m[0] = (double)nfree[k]; m[1] = (double)ncoll[k];
    m[2] = sx; m[3] = sy; m[4] = st; m[5] = sxx; m[6] = sxy; m[7] = sxt; m[8] = syy; m[9] = syt; m[10] = stt;
    free(free_pts[k]);
  }
  free(all_pts); free(all_hit); free(all_k);
  return 0;
}

/* truncateGMM's tail + the per-component EKF of the next step: MCSimulator.h:597-629, :766-771,
 * :804-812.  prev/next: K x ORC_STATE; moments: K x ORC_NMOM (global sums) or NULL at waypoint 0;
 * u, Md, z: applied control, diag(M), observations of the step leading to this waypoint. */
void orc_gmm_advance(const orc_config* cfg, const double* prev, const double* moments,
                     const double* u, const double* Md, const double* z, double* next) {
  int K = cfg->K;
  double raw[ORC_MAX_K] = {0};
  for (int k = 0; k < K; ++k) {
    const double* p = prev + k * ORC_STATE;
    double* o = next + k * ORC_STATE;
    memcpy(o, p, sizeof(double) * ORC_STATE);
    raw[k] = p[12];
    if (!moments) continue;
    raw[k] = 0.0;
    if (p[13] == 0.0) { o[12] = 0.0; continue; }
    const double* m = moments + k * ORC_NMOM;
    double n = m[0];
    if (!(n >= 2.0)) { o[13] = 0.0; o[12] = 0.0; continue; }            /* retired: build spec */
    double mean[3] = {m[2] / n, m[3] / n, m[4] / n};                     /* mean(free,1) :597 */
    double nm1 = n - 1.0;
    double cov[9];                                                       /* cov(free^T) :598 */
    cov[0] = (m[5] - (m[2] * m[2]) / n) / nm1;
    cov[1] = (m[6] - (m[2] * m[3]) / n) / nm1;
    cov[2] = (m[7] - (m[2] * m[4]) / n) / nm1;
    cov[4] = (m[8] - (m[3] * m[3]) / n) / nm1;
    cov[5] = (m[9] - (m[3] * m[4]) / n) / nm1;
    cov[8] = (m[10] - (m[4] * m[4]) / n) / nm1;
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    double pm[3], pc[9];
    orc_ekf_predict(mean, cov, u, Md, pm, pc);                           /* :769 */
    orc_ekf_update(pm, pc, z, cfg->L, cfg->lx, cfg->ly, cfg->Q);         /* :806 */
    memcpy(o, pm, sizeof pm);
    memcpy(o + 3, pc, sizeof pc);
    raw[k] = n;                                                          /* collisionCounts(1,k) :611 */
  }
  for (int k = 0; k < K; ++k) {                                          /* mvnrnd needs chol */
    double* o = next + k * ORC_STATE;
    double Ltmp[6];
    if (o[13] != 0.0 && !orc_chol3_lower(o + 3, Ltmp)) { o[13] = 0.0; raw[k] = 0.0; }
  }
  if (moments) {
    double w[ORC_MAX_K];
    orc_normalise_l1(raw, K, w);                                         /* :618-629 */
    for (int k = 0; k < K; ++k) next[k * ORC_STATE + 12] = w[k];
  } else {
    for (int k = 0; k < K; ++k) next[k * ORC_STATE + 12] = raw[k];
  }
}

void orc_gmm_initial_state(const orc_config* cfg, double* state) {     /* initGMM / initModel */
  int W = cfg->W;
  for (int k = 0; k < cfg->K; ++k) {
    double* s = state + k * ORC_STATE;
    memset(s, 0, sizeof(double) * ORC_STATE);
    s[0] = cfg->traj[0]; s[1] = cfg->traj[W]; s[2] = cfg->traj[2 * W];
    memcpy(s + 3, cfg->cov0, sizeof(double) * 9);
    s[12] = 1.0 / cfg->K;
    s[13] = 1.0;
  }
}

/* runGMMEstimation (MCSimulator.h:354-358 -> EKF_GaussProp("GMM") :649-864), all N samples.
 * probs_out[W], moments_out[W*K*11], states_out[W*K*16] optional.
 * last_samples (N x 3) / last_flags (N): the final waypoint's samples, optional. */
double orc_run_gmm(const orc_config* cfg, uint64_t seed, long long N, double* probs_out,
                   double* moments_out, double* states_out, double* last_samples,
                   int16_t* last_flags) {
  int W = cfg->W, K = cfg->K, L = cfg->L;
  size_t steps = (size_t)(W > 1 ? W - 1 : 1);
  double* applied = (double*)malloc(sizeof(double) * 3 * steps);
  double* Md = (double*)malloc(sizeof(double) * 3 * steps);
  double* z = (double*)malloc(sizeof(double) * (size_t)(L > 0 ? L : 1) * steps);
  orc_host_chain(cfg, seed, applied, Md, NULL, z, NULL, NULL);
  double state[ORC_MAX_K * ORC_STATE], next[ORC_MAX_K * ORC_STATE], mom[ORC_MAX_K * ORC_NMOM];
  orc_gmm_initial_state(cfg, state);
  orc_gmm_advance(cfg, state, NULL, NULL, NULL, NULL, next);
  memcpy(state, next, sizeof state);
  double prod = 1.0;
  for (int w = 0; w < W; ++w) {
    int last = (w == W - 1);
    if (states_out) memcpy(states_out + (size_t)w * K * ORC_STATE, state, sizeof(double) * K * ORC_STATE);
    orc_gmm_waypoint(cfg, seed, w, state, 0, N, N, mom, last ? last_samples : NULL,
                     last ? last_flags : NULL, NULL);
    if (moments_out) memcpy(moments_out + (size_t)w * K * ORC_NMOM, mom, sizeof(double) * K * ORC_NMOM);
    double collided = 0.0;
    for (int k = 0; k < K; ++k) collided += mom[k * ORC_NMOM + 1];
    double p = collided / (1.0 * (double)N);                             /* :639 */
    if (probs_out) probs_out[w] = p;
    prod *= (1.0 - p);                                                   /* :848-851 */
    if (!last) {
      orc_gmm_advance(cfg, state, mom, applied + 3 * w, Md + 3 * w, z + (size_t)L * w, next);
      memcpy(state, next, sizeof state);
    }
  }
  free(applied); free(Md); free(z);
  return 1.0 - prod;                                                     /* :856 */
}
