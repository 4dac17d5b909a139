"""Oracle primitives: known answers and accuracy against libm (CPU only)."""
import math

import numpy as np

# Random123 known-answer vectors for philox4x32-10 (kat_vectors of the Random123 distribution)
PHILOX_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def ulp_diff(a, b):
    return abs(a - b) / max(math.ulp(b), 5e-324)


# ... and for philox4x32-7 (same file of the Random123 distribution): the round count of the mixture-sample stream
PHILOX7_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]),
]


def test_philox_known_answers(orc):
    for ctr, key, want in PHILOX_KAT:
        assert orc.philox(ctr, key) == want
    for ctr, key, want in PHILOX7_KAT:
        assert orc.philox(ctr, key, rounds=7) == want


def test_log_within_one_ulp(orc):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(20000), 2.0 ** -rng.integers(1, 53, 2000) * rng.random(2000),
                         [1.0, 2.0 ** -53, 0.5, 0.70710678118654757, 0.7071067811865476, 1 - 2.0 ** -53]])
    worst = max(ulp_diff(orc.log(float(x)), math.log(float(x))) for x in xs if x > 0)
    assert worst <= 1.0, worst
    assert orc.log(1.0) == 0.0


def test_sincos_accuracy(orc):
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(-10, 10, 20000), rng.uniform(-1e4, 1e4, 5000),
                         [0.0, math.pi / 2, math.pi, 2 * math.pi, -math.pi / 4, 1e-300]])
    worst = 0.0
    for x in xs:
        s, c = orc.sincos(float(x))
        worst = max(worst, abs(s - math.sin(x)), abs(c - math.cos(x)))
        assert abs(s * s + c * c - 1.0) < 5e-16
    assert worst < 4e-16 * 1.5, worst      # absolute; |x| up to 1e4


def test_sincos_2pi_u32(orc):
    rng = np.random.default_rng(3)
    ws = np.concatenate([rng.integers(0, 2 ** 32, 20000), [0, 2 ** 29, 2 ** 30, 2 ** 31, 2 ** 32 - 1,
                                                            2 ** 29 - 1, 2 ** 29 + 1, 3 * 2 ** 29]])
    for w in ws:
        s, c = orc.sincos_2pi_u32(int(w))
        # reference through libm on the octant-reduced angle (2 pi w / 2^32 itself is not exact
        # in binary64, the reduced angle is to one rounding)
        q, m = int(w) >> 29, int(w) & (2 ** 29 - 1)
        phi = (q * 2 ** 29 + m) / 2.0 ** 29 * (math.pi / 4)
        phi_r = (m if q % 2 == 0 else 2 ** 29 - m) / 2.0 ** 29 * (math.pi / 4)
        rs, rc = math.sin(phi_r), math.cos(phi_r)
        want_s = [rs, rc, rc, rs, -rs, -rc, -rc, -rs][q]
        want_c = [rc, rs, -rs, -rc, -rc, -rs, rs, rc][q]
        assert abs(s - want_s) < 2.3e-16 and abs(c - want_c) < 2.3e-16, w
        assert abs(s - math.sin(phi)) < 1e-15 and abs(c - math.cos(phi)) < 1e-15, w
    assert orc.sincos_2pi_u32(0) == (0.0, 1.0)
    assert orc.sincos_2pi_u32(2 ** 30)[0] == 1.0
    assert abs(orc.sincos_2pi_u32(2 ** 30)[1]) < 1e-16


def test_normals_are_standard(orc):
    n = 40000
    z = np.array([orc.normal3(0x5EED0001, i, 7, 3)[0] for i in range(n)])
    assert np.all(np.isfinite(z))
    assert abs(z.mean()) < 4 / math.sqrt(3 * n)
    assert abs(z.var() - 1.0) < 0.02
    c = np.corrcoef(z.T)
    assert np.max(np.abs(c - np.eye(3))) < 0.02
    # tails: P(|z| > 3) = 0.0027
    frac = np.mean(np.abs(z) > 3)
    assert 0.0018 < frac < 0.0037
    # different waypoint / stream / seed => different draws; same arguments => same draws
    assert orc.normal3(1, 5, 0, 3) == orc.normal3(1, 5, 0, 3)
    assert orc.normal3(1, 5, 0, 3) != orc.normal3(1, 5, 1, 3)
    assert orc.normal3(1, 5, 0, 3) != orc.normal3(1, 5, 0, 2)
    assert orc.normal3(1, 5, 0, 3) != orc.normal3(2, 5, 0, 3)
    assert orc.normal3(1, 2 ** 32 + 5, 0, 3) != orc.normal3(1, 5, 0, 3)


def test_wrap_angle_keeps_two_pi(orc):
    # MCSimulator.h:56-65: the test is `> 2 pi`, so 2 pi itself stays; negatives go up
    tp = 2 * math.pi
    assert orc.wrap_angle(tp) == tp
    assert orc.wrap_angle(0.0) == 0.0
    assert abs(orc.wrap_angle(-0.01) - (tp - 0.01)) < 1e-15
    assert abs(orc.wrap_angle(tp + 0.25) - 0.25) < 1e-15
    assert abs(orc.wrap_angle(-3 * tp + 1.0) - 1.0) < 1e-14
    assert math.isnan(orc.wrap_angle(float("nan")))
