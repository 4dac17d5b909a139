"""C1, as far as it can be pinned: the collision predicate of this build is "the robot's footprint rectangle and an
obstacle rectangle overlap" (the reference asks OpenRAVE/ODE about the PR2 mesh instead, MCSimulator.h:279: not in its
tree, parity unpinned at that call -- DESIGN.md 8).  The oracle and the kernels decide it by separating axes with
folded margins; here the same question is answered by a DIFFERENT construction -- corners of one rectangle inside
the other, or two edges crossing -- on 20 000 random configurations (configurations within 1e-9 of touching are
skipped: the two constructions may round the boundary differently)."""
import numpy as np


def rect(cx, cy, hx, hy, ang):
    c, s = np.cos(ang), np.sin(ang)
    loc = np.array([[hx, hy], [-hx, hy], [-hx, -hy], [hx, -hy]])
    return np.array([cx, cy]) + loc @ np.array([[c, s], [-s, c]])


def overlap_by_corners_and_edges(A, B):
    """(overlap, margin): margin = the smallest |orientation determinant| met, scaled to a length."""
    margin = np.inf

    def inside(P, Q):                                  # any corner of P inside the convex quad Q (counter-clockwise)
        nonlocal margin
        hit = False
        for p in P:
            d = []
            for i in range(4):
                a, b = Q[i], Q[(i + 1) % 4]
                e = b - a
                d.append(((e[0] * (p[1] - a[1]) - e[1] * (p[0] - a[0])) / np.hypot(*e)))
            margin = min(margin, min(abs(v) for v in d))
            hit = hit or all(v >= 0 for v in d)
        return hit

    def cross(P, Q):
        nonlocal margin
        hit = False
        for i in range(4):
            p0, p1 = P[i], P[(i + 1) % 4]
            for j in range(4):
                q0, q1 = Q[j], Q[(j + 1) % 4]
                def side(a, b, c):
                    e = b - a
                    return (e[0] * (c[1] - a[1]) - e[1] * (c[0] - a[0])) / np.hypot(*e)
                s1, s2, s3, s4 = side(p0, p1, q0), side(p0, p1, q1), side(q0, q1, p0), side(q0, q1, p1)
                margin = min(margin, abs(s1), abs(s2), abs(s3), abs(s4))
                hit = hit or (s1 * s2 < 0 and s3 * s4 < 0)
        return hit

    return inside(A, B) or inside(B, A) or cross(A, B), margin


def test_separating_axes_against_corners_and_edges(orc):
    rng = np.random.default_rng(2024)
    n_hit = n_free = skipped = 0
    for _ in range(20000):
        fp = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.05, 0.6), rng.uniform(0.05, 0.6)]
        x, y, th = rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-7, 7)
        box = np.array([[rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.02, 1.5), rng.uniform(0.02, 1.5),
                         rng.choice([0.0, np.pi / 2, rng.uniform(-4, 4)])]])
        c, s = np.cos(th), np.sin(th)
        centre = np.array([x + c * fp[0] - s * fp[1], y + s * fp[0] + c * fp[1]])
        A = rect(centre[0], centre[1], fp[2], fp[3], th)
        B = rect(*box[0])
        want, margin = overlap_by_corners_and_edges(A, B)
        if margin < 1e-9:
            skipped += 1
            continue
        assert orc.collides(x, y, th, fp, box) == want, (x, y, th, fp, box, margin)
        n_hit += want
        n_free += not want
    assert n_hit > 3000 and n_free > 3000 and skipped < 50, (n_hit, n_free, skipped)
