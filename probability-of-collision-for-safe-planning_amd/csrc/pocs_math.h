// pocs_math.h -- numerics shared by the host chain and the HIP kernels (product code).
//
// Everything here is `__host__ __device__` and written so that the host build and the
// gfx950 build execute the *same sequence of IEEE-754 binary64 operations*:
//   * the translation units are compiled with -ffp-contract=off, every fused multiply-add
//     is an explicit fma();
//   * only +, -, *, /, sqrt, fma, rint and integer ops are used (all correctly rounded on
//     both sides), never a vendor libm transcendental.
// The spec ("POCS numerics v9") is written out in DESIGN.md section 4; the CPU oracle under
// oracle/ holds an independent plain-C restatement of the same spec and is never linked here.
//
// What this replaces in the reference: Armadillo's RNG + mvnrnd (GM_Model.h:83-116,
// MCSimulator.h:51-53,287-297; armadillo_bits/arma_rng.hpp:324-432) and libm cos/sin
// (MCSimulator.h:312-313,424-425).  The reference seeds from the clock/urandom
// (MCSimulator.h:141, GM_Model.h:53-54) so no stream of its own can be reproduced; ours is
// counter based (Philox4x32-10) so any (seed, index, waypoint) draw is a pure function.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define POCS_HD __host__ __device__ __forceinline__
#else
#define POCS_HD inline
#endif

// ----------------------------------------------------------------------------------------
// Philox4x32-R (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11).  Counter = 4 x u32, key = 2 x u32.  R = 10 rounds, Random123's default, for the host chain,
// the initial particle cloud and the component counts; R = 7 for the mixture samples, the stream the
// hot kernel draws per pair of samples: the paper's own Crush-resistant minimum for Philox4x32 (it passes
// BigCrush with 7 rounds; 10 is a safety margin), 30 % fewer of the integer multiplies that make up a
// fifth of the kernel.  Both round counts are pinned by Random123's known-answer vectors
// (tests/test_oracle_primitives.py).
// ----------------------------------------------------------------------------------------
struct pocs_u32x4 { uint32_t x, y, z, w; };

// a ^ b ^ c: one v_bitop3_b32 on gfx950 (the compiler emits two v_xor_b32 for the plain form)
POCS_HD uint32_t pocs_xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

template <int ROUNDS>
POCS_HD pocs_u32x4 pocs_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = pocs_xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = pocs_xor3((uint32_t)(p0 >> 32), c3, k1);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  pocs_u32x4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
  return o;
}

// Stream ids (counter word 3 = stream << 16 | slot).
#define POCS_STREAM_CHAIN 1u   // host chain: odometry + observation noise (MCSimulator.h:391-410,383-387)
#define POCS_STREAM_MCINIT 2u  // initial particle cloud                   (MCSimulator.h:287-297)
#define POCS_STREAM_GMM 3u     // mixture samples                          (GM_Model.h:83-116)

POCS_HD pocs_u32x4 pocs_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return pocs_philox4x32<10>(c0, c1, c2, c3, k0, k1);
}
#define POCS_GMM_PHILOX_ROUNDS 7

POCS_HD pocs_u32x4 pocs_draw(uint64_t seed, uint64_t index, uint32_t waypoint, uint32_t stream,
                             uint32_t slot) {
  if (stream == POCS_STREAM_GMM)          // (a literal at every call site: one of the two is compiled)
    return pocs_philox4x32<POCS_GMM_PHILOX_ROUNDS>((uint32_t)index, (uint32_t)(index >> 32), waypoint,
                                                   (stream << 16) | slot, (uint32_t)seed, (uint32_t)(seed >> 32));
  return pocs_philox4x32<10>((uint32_t)index, (uint32_t)(index >> 32), waypoint,
                             (stream << 16) | slot, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// ----------------------------------------------------------------------------------------
// log(x) for normal positive x.  Argument reduction x = 2^k (1+f), sqrt(1/2) < 1+f <= sqrt(2);
// s = f/(2+f); log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)), R = the classic degree-14 minimax
// polynomial in s (coefficients Lg1..Lg7 as published with Sun's fdlibm e_log.c).
// ----------------------------------------------------------------------------------------
POCS_HD double pocs_log(double x) {
  union { double d; uint64_t u; } b; b.d = x;
  uint32_t hx = (uint32_t)(b.u >> 32);
  int k = (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  const uint32_t i = (hx + 0x95f64u) & 0x100000u;   // 1+f > sqrt(2)  ->  halve, k += 1
  b.u = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (b.u & 0xffffffffull);
  k += (int)(i >> 20);
  const double f = b.d - 1.0;
  const double dk = (double)k;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
                            3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                   2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  // dk*ln2_hi - ((hfsq - (s*(hfsq+R) + dk*ln2_lo)) - f)
  return dk * 6.93147180369123816490e-01 -
         ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

// sin / cos on [-pi/4, pi/4] (degree-13 / degree-14 minimax kernels, coefficients as published
// with fdlibm k_sin.c / k_cos.c), Horner form on z = x^2 with explicit fma.
POCS_HD double pocs_ksin(double x) {
  const double z = x * x;
  double r = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  r = fma(z, r, 2.75573137070700676789e-06);
  r = fma(z, r, -1.98412698298579493134e-04);
  r = fma(z, r, 8.33333333332248946124e-03);
  r = fma(z, r, -1.66666666666666324348e-01);
  return fma(z * x, r, x);
}
POCS_HD double pocs_kcos(double x) {
  const double z = x * x;
  double r = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  r = fma(z, r, -2.75573143513906633035e-07);
  r = fma(z, r, 2.48015872894767294178e-05);
  r = fma(z, r, -1.38888888888741095749e-03);
  r = fma(z, r, 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * z) * r);
}

// sin and cos of an arbitrary angle |x| < 2^20: three-step Cody-Waite reduction by pi/2
// (33+33+53 bit split of pi/2), then the kernels above.  Absolute error < 2 ulp(1).
POCS_HD void pocs_sincos(double x, double* sn, double* cs) {
  const double fn = rint(x * 6.36619772367581382433e-01);
  const int n = (int)fn;
  double r = x - fn * 1.57079632673412561417e+00;
  r = r - fn * 6.07710050630396597660e-11;
  r = r - fn * 2.02226624879595063154e-21;
  const double ks = pocs_ksin(r), kc = pocs_kcos(r);
  const double a = (n & 1) ? kc : ks;     // |sin|-like term
  const double b = (n & 1) ? ks : kc;     // |cos|-like term
  *sn = (n & 2) ? -a : a;
  *cs = ((n + 1) & 2) ? -b : b;
}

// sin and cos of 2*pi*t for t = w * 2^-32 given as the 32-bit integer w: the top three bits
// select the octant, the low 29 bits give the position inside it; no range reduction error.
POCS_HD void pocs_sincos_2pi_u32(uint32_t w, double* sn, double* cs) {
  const uint32_t q = w >> 29;
  const uint32_t m = w & 0x1fffffffu;
  const uint32_t mm = (q & 1u) ? (0x20000000u - m) : m;            // odd octant: mirror
  const double phi = ((double)mm * 0x1p-29) * 7.85398163397448279e-01;   // in [0, pi/4]
  const double ks = pocs_ksin(phi), kc = pocs_kcos(phi);
  const bool swap = ((q + 1u) & 2u) != 0u;                          // octants 1,2,5,6
  const double a = swap ? kc : ks;                                  // |sin|
  const double b = swap ? ks : kc;                                  // |cos|
  *sn = (q & 4u) ? -a : a;                                          // octants 4..7
  *cs = ((q + 2u) & 4u) ? -b : b;                                   // octants 2..5
}

// One Box-Muller pair from three words of a Philox draw:
//   u = (((w1:w0) >> 11) + 1) * 2^-53 in (0,1],  radius = sqrt(-2 log u),  angle = 2 pi w2 2^-32.
POCS_HD void pocs_normal_pair(uint32_t w0, uint32_t w1, uint32_t w2, double* n0, double* n1) {
  const uint64_t a = ((((uint64_t)w1) << 32) | (uint64_t)w0) >> 11;
  const double u = (double)(a + 1ull) * 0x1p-53;
  const double rad = sqrt(-2.0 * pocs_log(u));
  double sn, cs;
  pocs_sincos_2pi_u32(w2, &sn, &cs);
  *n0 = rad * cs;
  *n1 = rad * sn;
}

// ----------------------------------------------------------------------------------------
// Table-driven forms used on the hot path (mixture samples, footprint heading): less than half the
// instructions of the polynomial forms above.  The 12 KB of tables are built once on the host FROM THE
// FUNCTIONS ABOVE (so product and oracle, whose functions agree bit for bit, build identical tables) and
// staged in LDS by the kernels.  Numerics v9 (tables as in v8):
//   lg[i] = {invc_i, 2 log(invc_i) + 2 ln 2},  c_i = 1 + (i + 1/2)/512,  invc_i = 1/c_i rounded
//   sc[s] = {cos, sin} of 2 pi s / 256: the sector BOUNDARIES
// ----------------------------------------------------------------------------------------
struct pocs_tables {
  double lg[512][2];
  double sc[256][2];
};

// Polynomial constants of the forms below (v9).  BM_*: sine and cosine of the Box-Muller angle IN UNITS OF THE
// WORD'S LOW 24 BITS read as a signed number, k in [-2^23, 2^23) an exact integer, angle d = k a with a = 2 pi 2^-32:
//   sin d = k (a - a^3/6 k^2 + a^5/120 k^4),   cos d = 1 - a^2/2 k^2 + C4 a^4 k^4
// (v8 multiplied k by a first and ran the polynomials in d: one product more per angle).  C4 is not 1/24: the cosine
// stops at d^4 (v8: d^6) and C4 is the coefficient that minimises the largest error of 1 - d^2/2 + C4 d^4 over
// |d| <= pi/256 with the first two coefficients held at the instruction set's inline constants 1 and -1/2:
// 5.0e-16 at most (Taylor's 1/24: 4.7e-15; computed with 60 digits, tools/make_v9_constants.py).  The sine
// keeps d^5: truncation 8.3e-18.
#define POCS_BM_S1 1.462918079267159624024e-09    /* a             0x1.921fb54442d18p-30 */
#define POCS_BM_S3 -5.218056424438286096208e-28   /* -a^3 / 6      -0x1.4abbce625be53p-91 */
#define POCS_BM_S5 5.583657738838274755966e-47    /* a^5 / 120     0x1.466bc6775aae2p-154 */
#define POCS_BM_C2 -1.070064653323357779997e-18   /* -a^2 / 2      -0x1.3bd3cc9be45dep-60 */
#define POCS_BM_C4 1.908388704914255060734e-37    /* C4 a^4        0x1.03c1a4196664ep-122 */
#define POCS_COS_C4 4.166647965169937434249e-02   /* C4            0x1.5554f0ee31235p-5 */
#define POCS_2LN2 1.386294361119890572454e+00

// The first Horner step of each polynomial multiplies by one constant and adds another; a gfx950 VALU instruction
// reads at most one literal / scalar operand, so one of the two needs a register and the compiler materialises it
// with a v_mov_b64 EVERY time.  The hot kernel keeps the three addends that are not inline constants in vector
// registers for the length of its loop instead (POCS_VCONST pins them there); same values, same operations.
// Everyone else passes nullptr and gets the literals.  (The radius polynomial and the heading's cosine need none:
// their addends are inline constants of the instruction set.)
struct pocs_vconst { double bm_s3, bm_c2, sin_c3; };
#if defined(__HIP_DEVICE_COMPILE__)
#define POCS_VCONST(name) pocs_vconst name = {POCS_BM_S3, POCS_BM_C2, -1.0 / 6.0}; \
  asm volatile("" : "+v"(name.bm_s3), "+v"(name.bm_c2), "+v"(name.sin_c3))
#else
#define POCS_VCONST(name) pocs_vconst name = {POCS_BM_S3, POCS_BM_C2, -1.0 / 6.0}
#endif

POCS_HD void pocs_tables_init(pocs_tables* T) {
  for (int i = 0; i < 512; ++i) {
    const double c = 1.0 + ((double)i + 0.5) * 0x1p-9;
    const double invc = 1.0 / c;
    T->lg[i][0] = invc;
    T->lg[i][1] = 2.0 * pocs_log(invc) + POCS_2LN2;
  }
  for (int s = 0; s < 256; ++s) {
    double sn, cs;
    pocs_sincos_2pi_u32((uint32_t)s << 24, &sn, &cs);
    T->sc[s][0] = cs;
    T->sc[s][1] = sn;
  }
}

// The squared Box-Muller radius -2 log(w 2^-32) of a 32-bit word w, i.e. of a uniform with the 2^32 levels
// 0, 2^-32, ..., 1 - 2^-32, in ONE pass:
//   w = 2^e t, t in [1, 2), e = 31 - z with z the number of leading zero bits of w; i = top 9 mantissa bits;
//   r = t invc_i - 1 (one fma, |r| <= 2^-10);
//   -2 log t = 2 log(invc_i) - 2 log1p(r),   -2 log1p(r) = r (-2 + r (1 + r (-2/3 + r/2)))   (truncation < 4e-16);
//   -2 log(2^(e-32)) = (z + 1) 2 ln 2, the "+ 1" of which the table entry already holds:
//   result = fma(r, p(r), fma(z, 2 ln 2, lg[i][1]))
// (v8: the word + 1, the exponent out of frexp, the three terms added in two steps and an fma: one addition, one
// subtraction and one product more).  The inner fma is rounded once at its own magnitude: where its terms are small
// (u near 1: z = 0, the table entry 2 log(2 invc_i) itself, near 0) it carries the 1e-16 of the entry, six orders below
// the 4.7e-10 between two neighbouring levels.  Absolute error <= 1e-15 over all words (CPU tests).
// THE LEVEL 0 (one word in 2^32) has no logarithm; it is given the value the same instructions produce from what the
// hardware returns for a zero word -- mantissa 0, hence cell 0 and r = -1; z = -1 (v_ffbh_u32 finds no bit) --:
// 25/6 + 2 log(invc_0) = 4.1647, a radius of 2.0408: finite and nowhere near the bound the obstacle culling of
// k_gmm_step works with, which the level 2^-32 sets: sqrt(64 ln 2) < 6.661.  The other end, u = 1 - 2^-32: 4.66e-10.
POCS_HD double pocs_radius2_unit32(uint32_t w, const pocs_tables* T) {
#if defined(__HIP_DEVICE_COMPILE__)
  // w = 2^(e+1) mant with mant = t / 2 in [1/2, 1): the hardware's frexp gives mant in one instruction; the table
  // entry's byte offset is a shift and a mask of the high word; and r = fma(t, invc, -1) = fma(mant, 2 invc, -1)
  // exactly, for which the kernel's LDS copy of the table holds 2 invc (stage_tables doubles the entry on its way
  // in: exact).  The leading zeros: v_ffbh_u32, which returns -1 for a zero word (__builtin_clz leaves that open).
  union { double d; uint64_t u; } b; b.d = (double)w;
  int lz;
  asm("v_ffbh_u32 %0, %1" : "=v"(lz) : "v"(w));
  const double mant = __builtin_amdgcn_frexp_mant(b.d);           // (0 for w = 0)
  const unsigned off = ((unsigned)(b.u >> 32) >> 7) & 0x1ff0u;    // 16 i
  const double* ent = reinterpret_cast<const double*>(reinterpret_cast<const char*>(&T->lg[0][0]) + off);
  const double r = fma(mant, ent[0], -1.0);
  const double l2c = ent[1];
#else
  double r, l2c;
  int lz;
  if (w == 0u) {
    r = -1.0; l2c = T->lg[0][1]; lz = -1;
  } else {
    union { double d; uint64_t u; } b; b.d = (double)w;
    const int i = (int)(b.u >> 43) & 511;
    b.u = (b.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;   // t
    r = fma(b.d, T->lg[i][0], -1.0);
    l2c = T->lg[i][1];
    lz = __builtin_clz(w);
  }
#endif
  double p = fma(r, 0.5, -2.0 / 3.0);
  p = fma(r, p, 1.0);
  p = fma(r, p, -2.0);
  return fma(r, p, fma((double)lz, POCS_2LN2, l2c));
}

// sin / cos of a small angle |d| <= pi/256 in radians (the heading's remainder): sine to d^5 (truncation < 1e-17),
// cosine to d^4 with the fitted coefficient above (error <= 5.0e-16)
POCS_HD void pocs_sincos_small(double d, double* sd, double* cd, const pocs_vconst* V = nullptr) {
  const double z = d * d;
  const double ps = fma(z, 1.0 / 120.0, V ? V->sin_c3 : -1.0 / 6.0);
  *sd = fma(d * z, ps, d);
  const double pc = fma(z, POCS_COS_C4, -0.5);
  *cd = fma(z, pc, 1.0);
}

// sin and cos of the Box-Muller angle of a 32-bit word: sector s = top 8 bits, k = the low 24 bits READ AS A SIGNED
// NUMBER in [-2^23, 2^23) (one v_bfe_i32; v8 masked and subtracted 2^23), an offset from that sector's table entry:
// the angle 2 pi (2^24 s + k) 2^-32.  Words <-> the 2^32 equally spaced angles is one to one (words with bit 23 set
// land in the half sector BELOW their boundary, the others above), so the angle is uniform on those levels like
// 2 pi w 2^-32 itself.  The polynomials run in k (above): k and k^2 are exact.
POCS_HD void pocs_sincos_2pi_u32_tab(uint32_t w, const pocs_tables* T, double* sn, double* cs, const pocs_vconst* V = nullptr) {
  const int s = (int)(w >> 24);
  const double k = (double)((int32_t)(w << 8) >> 8);                      // sign-extended low 24 bits: [-2^23, 2^23)
  const double z = k * k;
  double ps = fma(z, POCS_BM_S5, V ? V->bm_s3 : POCS_BM_S3);
  ps = fma(z, ps, POCS_BM_S1);
  const double sd = k * ps;
  const double pc = fma(z, POCS_BM_C4, V ? V->bm_c2 : POCS_BM_C2);
  const double cd = fma(z, pc, 1.0);
  const double C = T->sc[s][0], S = T->sc[s][1];
  *sn = fma(S, cd, C * sd);
  *cs = fma(C, cd, -(S * sd));
}

// sin and cos of an arbitrary angle by the same sectors: n = rint(x * 256/(2 pi)), d = x - n P in ONE fma with
// P = pi/128 rounded to double (v8: two Cody-Waite steps): d carries n (P - pi/128), at most |x| * 4e-17 -- a
// heading of a few turns is off by a few 1e-16 rad, what its own last bit is worth --, so |d| <= pi/256 (1 + 1e-15)
// from the table entry of sector n mod 256.  Arguments: |x| < 2^20 (n must fit an int with room to spare).
POCS_HD void pocs_sincos_tab(double x, const pocs_tables* T, double* sn, double* cs, const pocs_vconst* V = nullptr) {
  const double fn = rint(x * 4.07436654315252084757e+01);
  const int n = (int)fn;
  const double d = fma(-fn, 2.45436926061702587187e-02, x);
  double sd, cd;
  pocs_sincos_small(d, &sd, &cd, V);
  const int s = n & 255;
  const double C = T->sc[s][0], S = T->sc[s][1];
  *sn = fma(S, cd, C * sd);
  *cs = fma(C, cd, -(S * sd));
}

// The three standard normals (+ one spare uniform word) that belong to (seed, index, waypoint)
// on a stream: slot 0 -> z0, z1 and the spare word; slot 1 -> z2.
POCS_HD void pocs_normal3(uint64_t seed, uint64_t index, uint32_t waypoint, uint32_t stream,
                          double z[3], uint32_t* spare) {
  const pocs_u32x4 a = pocs_draw(seed, index, waypoint, stream, 0u);
  const pocs_u32x4 b = pocs_draw(seed, index, waypoint, stream, 1u);
  double unused;
  pocs_normal_pair(a.x, a.y, a.z, &z[0], &z[1]);
  pocs_normal_pair(b.x, b.y, b.z, &z[2], &unused);
  *spare = a.w;
}

// sqrt for positive t in the normal range -- the squared Box-Muller radius (0 < t <= 64), and the distances and pivots
// of pocs_model.h --, correctly rounded like sqrt().  On the
// device: the compiler's own f64 expansion (v_rsq_f64 seed, one coupled Goldschmidt step, two
// residual corrections) without the 2^+-256 range scaling it wraps around it for arguments below
// 2^-767 and without its select for +-0 / inf: t = |-2 log u| is never 0 (the smallest value, at
// u = 1, is a rounding error of the table form, ~1e-16; every other word gives >= 4.6e-10 --
// pinned by tests/test_product_host_vs_oracle.py) and never exceeds 44.4.
POCS_HD double pocs_sqrt_normal(double t) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(t);
  double g = t * y;
  double h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, t);
  g = fma(d, h, g);
  d = fma(-g, g, t);
  g = fma(d, h, g);
  return g;
#else
  return sqrt(t);
#endif
}

// Division and square root of the O(1)-per-waypoint estimator math (pocs_model.h: truncated moments, EKF update,
// Cholesky) as one wave executes them between two sampling launches, where every instruction of the serial chain is
// exposed latency: the compiler's own IEEE expansions -- v_rcp_f64 / v_rsq_f64 seed, Newton steps, one or two residual
// corrections -- WITHOUT the v_div_scale / v_div_fmas / v_div_fixup wrapping (the 2^+-256 scaling for sqrt) that
// guards operands near the ends of the exponent range and infinities: covariances, distances and counts are nowhere
// near them, and for such operands the guarded and the bare sequence execute the same arithmetic on the same values
// -- the correctly rounded quotient / root that `/` and sqrt() give on the host.  pocs_recip_seed is the refined
// reciprocal a group of quotients with one denominator shares.
POCS_HD double pocs_recip_seed(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(b);
  double e = fma(-b, y, 1.0);
  y = fma(y, e, y);
  e = fma(-b, y, 1.0);
  return fma(y, e, y);
#else
  return b;                                        // (host: the denominator itself; pocs_div_by then divides)
#endif
}
POCS_HD double pocs_div_by(double a, double b, double seed) {      // a / b; seed = pocs_recip_seed(b)
#if defined(__HIP_DEVICE_COMPILE__)
  const double q = a * seed;
  const double r = fma(-b, q, a);
  return fma(r, seed, q);
#else
  (void)seed;
  return a / b;
#endif
}
POCS_HD double pocs_div(double a, double b) { return pocs_div_by(a, b, pocs_recip_seed(b)); }

// Box-Muller pair of the mixture sampler, through the tables: one word for the radius,
// u = wr 2^-32, radius = sqrt(-2 log u) <= sqrt(64 ln 2) < 6.661 (the bound the obstacle culling of
// k_gmm_step relies on; the level u = 0: pocs_radius2_unit32), one word for the angle.
POCS_HD void pocs_normal_pair_w2(uint32_t wr, uint32_t wa, const pocs_tables* T, double* n0, double* n1, const pocs_vconst* V = nullptr) {
  // (|.|: free on the device, and the root's argument is then non-negative by construction, not by an error bound)
  const double rad = pocs_sqrt_normal(fabs(pocs_radius2_unit32(wr, T)));
  double sn, cs;
  pocs_sincos_2pi_u32_tab(wa, T, &sn, &cs, V);
  *n0 = rad * cs;
  *n1 = rad * sn;
}

// The six standard normals and two spare words of a PAIR of mixture samples (2j, 2j+1), from
// TWO draws keyed by the pair index j -- every word is used, no Box-Muller output is thrown away:
//   slot 0: (x, y) -> z0, z1 of sample 2j      (z, w) -> z2 of sample 2j, z0 of sample 2j+1
//   slot 1: (x, y) -> z1, z2 of sample 2j+1    z, w  -> the spare words of samples 2j, 2j+1
POCS_HD void pocs_normal3_pair(uint64_t seed, uint64_t pair, uint32_t waypoint, uint32_t stream,
                               const pocs_tables* T, double za[3], double zb[3], uint32_t* spare_a,
                               uint32_t* spare_b, const pocs_vconst* V = nullptr) {
  const pocs_u32x4 a = pocs_draw(seed, pair, waypoint, stream, 0u);
  const pocs_u32x4 b = pocs_draw(seed, pair, waypoint, stream, 1u);
  pocs_normal_pair_w2(a.x, a.y, T, &za[0], &za[1], V);
  pocs_normal_pair_w2(a.z, a.w, T, &za[2], &zb[0], V);
  pocs_normal_pair_w2(b.x, b.y, T, &zb[1], &zb[2], V);
  *spare_a = b.z;
  *spare_b = b.w;
}
