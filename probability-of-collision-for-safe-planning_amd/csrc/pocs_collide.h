// pocs_collide.h -- the collision predicate (product code, host + device).
//
// Stands in for MCSimulator::checkCollision (MCSimulator.h:257-285), i.e. for
// robot->SetActiveDOFValues + env->CheckCollision(robot) (:275,:279).  OpenRAVE, its ODE checker
// and the PR2 model are not part of the reference tree, so this is the build's own explicit 2-D
// model: one oriented footprint box carried by the base pose (x, y, theta) against M static
// oriented boxes, separating-axis test on the four face normals.  Touching counts as collision.
//
// Obstacle record (POCS_OBS_STRIDE doubles), prepared once on the host by pocs_prepare_obstacle:
//   cx cy   centre            ax ay   unit x-axis of the box (cos yaw, sin yaw)
//   hx hy   half extents      bx by   half extents of its world AABB grown by the footprint's
//                                     bounding radius (broad phase; purely conservative)
#pragma once
#include "pocs_math.h"

#define POCS_OBS_STRIDE 8
#define POCS_MAX_OBSTACLES 64

struct pocs_footprint { double dx, dy, hx, hy; };   // offset in the base frame, half extents

POCS_HD void pocs_prepare_obstacle(const double box5[5], const pocs_footprint* fp, double* rec) {
  double sn, cs;
  pocs_sincos(box5[4], &sn, &cs);
  const double rr = sqrt(fp->hx * fp->hx + fp->hy * fp->hy);
  rec[0] = box5[0]; rec[1] = box5[1];
  rec[2] = cs; rec[3] = sn;
  rec[4] = box5[2]; rec[5] = box5[3];
  rec[6] = (box5[2] * fabs(cs) + box5[3] * fabs(sn)) + rr;
  rec[7] = (box5[2] * fabs(sn) + box5[3] * fabs(cs)) + rr;
}

// Narrow phase: footprint at (px, py) with heading (sn, cs) against one obstacle record that has
// passed the broad phase.  Separating axes = the four face normals; a > b <=> a - b > 0 exactly
// in IEEE arithmetic (gradual underflow), so each pair of tests folds into one compare of the larger
// margin.  An axis-aligned obstacle (axis exactly (1, 0)) takes a shorter path that produces the
// same values: fma(c, 1, s*0) == c, fma(dx, 1, dy*0) == dx.
// The margins of the four axes given the relative yaw (cr, sr: cosine and sine, signed) and the offset in the
// obstacle's frame (e1, e2).  Inlined into BOTH paths of pocs_box_narrow below instead of after their merge: merged,
// the compiler materialises |cos|, |sin| and copies of the offsets into fresh registers for the common tail (six
// vector moves per pose and record in the axis-aligned case); in place, the absolute values are operand modifiers.
POCS_HD bool pocs_box_margins(double dx, double dy, double sn, double cs, double rx, double ry, double hx, double hy,
                              double cr, double sr, double e1, double e2) {
  // The obstacle's own two axes first: they separate nearly every pose that is in reach but does not
  // touch (a wall and a robot driving past it), and a wave whose poses are all separated there skips
  // the footprint's axes.  (max of the four margins > 0  <=>  one of the two maxima > 0.)
  const double m3 = fabs(e1) - (hx + fma(rx, fabs(cr), ry * fabs(sr)));
  const double m4 = fabs(e2) - (hy + fma(rx, fabs(sr), ry * fabs(cr)));
  if (fmax(m3, m4) > 0.0) return false;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("; footprint axes");                                // keeps the early return a branch
#endif
  const double d1 = fma(dx, cs, dy * sn);                         // d in the footprint frame
  const double d2 = fma(dy, cs, -(dx * sn));
  const double m1 = fabs(d1) - (rx + fma(hx, fabs(cr), hy * fabs(sr)));
  const double m2 = fabs(d2) - (ry + fma(hx, fabs(sr), hy * fabs(cr)));
  return !(fmax(m1, m2) > 0.0);
}

POCS_HD bool pocs_box_narrow(double px, double py, double sn, double cs, double rx, double ry,
                             const double* o) {
  const double dx = o[0] - px;
  const double dy = o[1] - py;
  const double ax = o[2], ay = o[3], hx = o[4], hy = o[5];
  if (ax == 1.0 && ay == 0.0)                                     // wave-uniform: the relative yaw IS the heading, the offset as it is
    return pocs_box_margins(dx, dy, sn, cs, rx, ry, hx, hy, cs, sn, dx, dy);
  return pocs_box_margins(dx, dy, sn, cs, rx, ry, hx, hy,
                          fma(cs, ax, sn * ay),                   // cos of the relative yaw
                          fma(sn, ax, -(cs * ay)),                // sin of the relative yaw
                          fma(dx, ax, dy * ay),                   // d in the obstacle frame
                          fma(dy, ax, -(dx * ay)));
}

// Broad phase + narrow phase against one record (host-side convenience, same result).
POCS_HD bool pocs_box_hit(double px, double py, double sn, double cs, double rx, double ry,
                          const double* o) {
  if (fabs(o[0] - px) > o[6] || fabs(o[1] - py) > o[7]) return false;
  return pocs_box_narrow(px, py, sn, cs, rx, ry, o);
}

// checkCollision for one pose: true if the footprint touches any of the M obstacles.
POCS_HD bool pocs_pose_collides(double x, double y, double th, const pocs_footprint* fp,
                                const double* obs, int M, const pocs_tables* T, const pocs_vconst* V = nullptr) {
  if (M <= 0) return false;                     // nothing in reach (k_gmm_step: every obstacle culled): no heading needed
  double sn, cs;
  pocs_sincos_tab(th, T, &sn, &cs, V);
  double px = x, py = y;
  if (!(fp->dx == 0.0 && fp->dy == 0.0)) {      // a centred footprint skips x + (c*0 - s*0) == x
#if defined(__HIP_DEVICE_COMPILE__)
    // keeps this a (scalar) BRANCH: left alone the compiler computes both and selects, 10 vector
    // instructions per pose for a footprint that is centred in every scene of the reference
    asm volatile("; offset footprint");
#endif
    px = x + fma(cs, fp->dx, -(sn * fp->dy));
    py = y + fma(sn, fp->dx, cs * fp->dy);
  }
  bool hit = false;                             // (set under a condition, not or-ed in: see pocs_pair_collides)
  for (int m = 0; m < M; ++m)
    if (pocs_box_hit(px, py, sn, cs, fp->hx, fp->hy, obs + m * POCS_OBS_STRIDE)) hit = true;
  return hit;
}

// The same predicate for the two poses a thread of k_gmm_step draws per iteration, with ONE pass over the obstacle
// table: every record is read once for both poses (the table sits in LDS), the loop's bookkeeping is paid once.
// Per pose exactly the operations of pocs_pose_collides, in the same order: the same flags.
//   EAGER (k_gmm_step's lone form, whose sampling phase is latency): a record's eight doubles are all requested at the
//   head of its iteration and waited for once, instead of field by field as the tests get to them (three LDS round
//   trips per pose where lanes reach the narrow phase)
template <bool EAGER = false>
POCS_HD void pocs_pair_collides(const double x[2], const double y[2], const double th[2], const pocs_footprint* fp,
                                const double* obs, int M, const pocs_tables* T, const pocs_vconst* V, bool hit[2]) {
  hit[0] = false; hit[1] = false;
  if (M <= 0) return;
  double sn[2], cs[2], px[2], py[2];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int h = 0; h < 2; ++h) {
    pocs_sincos_tab(th[h], T, &sn[h], &cs[h], V);
    px[h] = x[h]; py[h] = y[h];
  }
  if (!(fp->dx == 0.0 && fp->dy == 0.0)) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("; offset footprint");
#endif
    for (int h = 0; h < 2; ++h) {
      px[h] = x[h] + fma(cs[h], fp->dx, -(sn[h] * fp->dy));
      py[h] = y[h] + fma(sn[h], fp->dx, cs[h] * fp->dy);
    }
  }
  for (int m = 0; m < M; ++m) {
    const double* o = obs + m * POCS_OBS_STRIDE;
#if defined(__HIP_DEVICE_COMPILE__)
    double rec[POCS_OBS_STRIDE];
    if (EAGER) {
#pragma unroll
      for (int q = 0; q < POCS_OBS_STRIDE; ++q) rec[q] = o[q];
      asm volatile("" : "+v"(rec[0]), "+v"(rec[1]), "+v"(rec[2]), "+v"(rec[3]), "+v"(rec[4]), "+v"(rec[5]), "+v"(rec[6]), "+v"(rec[7]));
      o = rec;
    }
#pragma unroll
#endif
    // `if (...) hit = true`, not `hit |= ...`: the flag then stays a lane mask in scalar registers across the
    // loop, merged by scalar instructions; or-ed in as an integer it cost a v_cndmask and a v_or per record and pose
    for (int h = 0; h < 2; ++h)
      if (pocs_box_hit(px[h], py[h], sn[h], cs[h], fp->hx, fp->hy, o)) hit[h] = true;
  }
}

// The footprint's largest half-extent along world x over all headings in [lo, hi] (along world y: the
// same function of [lo - pi/2, hi - pi/2]): an upper bound, never more than the bounding radius.
//   f(t) = rx |cos t| + ry |sin t| is concave between the multiples of pi/2 and peaks with the bounding
//   radius at t = +-atan(ry / rx) + k pi: with no peak inside the range the maximum sits at an end.
// k_gmm_step's culling uses it to give the records it keeps a broad phase that fits the task's headings
// (gmm_cull); tests/test_product_host_vs_oracle.py scans f densely against it.
//   rr = the bounding radius sqrt(rx^2 + ry^2), phi = atan2(ry, rx): constants of the footprint, computed once on the
//   host for the kernels (every block of every launch culls: a square root, an arc tangent and four divisions less
//   in each of them)
// (in three steps, so that a wave can evaluate the end values of several ranges side by side, one per lane: gmm_cull)
//   is the bounding radius the answer without looking at the ends (a range of pi or more, a peak inside it)?
POCS_HD bool pocs_footprint_extent_is_radius(double phi, double lo, double hi) {
  const double PI = 3.14159265358979323846, INV_PI = 0.318309886183790671538;
  if (!(hi - lo < PI)) return true;
  for (int sgn = -1; sgn <= 1; sgn += 2) {
    const double s = sgn * phi;
    if (ceil((lo - s) * INV_PI) <= floor((hi - s) * INV_PI)) return true;   // a peak inside the range
  }
  return false;
}
//   f at one end of a range
POCS_HD double pocs_footprint_extent_end(double rx, double ry, double t) {
  double sn, cs;
  pocs_sincos(t, &sn, &cs);
  return fma(rx, fabs(cs), ry * fabs(sn));
}
//   the bound from the two end values
POCS_HD double pocs_footprint_extent_of_ends(double rr, double fa, double fb) {
  return fmin(rr, fmax(fa, fb) * (1.0 + 1e-9) + 1e-12);
}
POCS_HD double pocs_footprint_extent_pre(double rx, double ry, double rr, double phi, double lo, double hi) {
  if (pocs_footprint_extent_is_radius(phi, lo, hi)) return rr;
  const double fa = pocs_footprint_extent_end(rx, ry, lo);
  const double fb = pocs_footprint_extent_end(rx, ry, hi);
  return pocs_footprint_extent_of_ends(rr, fa, fb);
}
POCS_HD double pocs_footprint_extent(double rx, double ry, double lo, double hi) {
  return pocs_footprint_extent_pre(rx, ry, sqrt(rx * rx + ry * ry), atan2(ry, rx), lo, hi);
}
