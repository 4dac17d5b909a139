"""The polynomial constants of numerics v9 (csrc/pocs_math.h, oracle/pocs_oracle.c) with 60 digits: the scaled
Box-Muller coefficients and the cosine's fitted d^4 coefficient.  usage: python3 tools/make_v9_constants.py"""
import mpmath as mp, struct
mp.mp.dps = 60
def dbl(x): return float(x)       # mp -> nearest double
def show(name, x):
    d = dbl(x); print("%-6s %.21e  %s  relerr %.2e" % (name, d, d.hex(), float(abs(mp.mpf(d)-x)/abs(x))))
    return d
Z = (mp.pi/256)**2 * (1 + mp.mpf(2)**-20)
# minimax c4 for cos(sqrt z) ~ 1 - z/2 + c4 z^2 on [0, Z]
def err(c4, z): return mp.cos(mp.sqrt(z)) - (1 - z/2 + c4*z*z)
def maxerr(c4):
    best = 0
    for i in range(1, 2001):
        z = Z*i/2000
        best = max(best, abs(err(c4, z)))
    return best
lo, hi = mp.mpf(1)/24 - mp.mpf('4e-7'), mp.mpf(1)/24
for _ in range(80):
    m1 = lo + (hi-lo)/3; m2 = hi - (hi-lo)/3
    if maxerr(m1) < maxerr(m2): hi = m2
    else: lo = m1
c4 = (lo+hi)/2
print("c4 minimax", c4, "max err", maxerr(c4), " taylor err", maxerr(mp.mpf(1)/24))
C4 = show("C4", c4)
print("maxerr with rounded", maxerr(mp.mpf(C4)))
a = 2*mp.pi/mp.mpf(2)**32
show("S1", a); show("S3", -a**3/6); show("S5", a**5/120); show("C2s", -a*a/2); show("C4s", mp.mpf(C4)*a**4)
show("P", mp.pi/128); show("INVP", 128/mp.pi)
# sin truncation
d = mp.pi/256
print("sin trunc", abs(mp.sin(d) - (d - d**3/6 + d**5/120)))
