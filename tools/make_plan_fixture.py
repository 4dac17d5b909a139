#!/usr/bin/env python3
"""Re-encode the reference's bundled plan (trajectory.dat / odometry.dat) as plain text.

The two files are Python-2 pickles (protocol 0, i.e. *text*).  Nothing is unpickled
here: the files are read as text and only two token kinds are decoded by hand --
`S'...'` string literals whose decoded length is 8 or 24 bytes (little-endian float64
payloads of numpy scalars / 3-vectors) and `F<repr>` float lines (the goal row of
trajectory.dat holds plain floats).  No opcode is executed, no class is resolved.

Reference: hw2_astar.py:195-204 writes the files, MCSimulation.py:176-198 reads them.
Run only where /root/reference exists; the outputs are committed.

Usage: python tools/make_plan_fixture.py [/root/reference] [out_dir]
"""
import re
import struct
import sys
from pathlib import Path

_ESC = {ord("n"): 10, ord("t"): 9, ord("r"): 13, ord("\\"): 92, ord("'"): 39, ord('"'): 34,
        ord("a"): 7, ord("b"): 8, ord("f"): 12, ord("v"): 11, ord("0"): 0}


def _unescape_py2_repr(body: bytes) -> bytes:
    out = bytearray()
    i = 0
    while i < len(body):
        c = body[i]
        if c != 0x5C:
            out.append(c)
            i += 1
            continue
        n = body[i + 1]
        if n == ord("x"):
            out.append(int(body[i + 2:i + 4].decode("ascii"), 16))
            i += 4
        elif n in _ESC:
            out.append(_ESC[n])
            i += 2
        else:  # unknown escape: python keeps the backslash
            out.append(c)
            i += 1
    return bytes(out)


def decode_doubles(path: Path):
    vals = []
    for line in path.read_bytes().split(b"\n"):
        line = line.lstrip(b"a")            # a leading APPEND opcode may share the line
        m = re.match(rb"[a-z]*S(['\"])(.*)\1$", line)   # e.g. "tbS'...'" (TUPLE, BUILD, then STRING)
        if m:
            raw = _unescape_py2_repr(m.group(2))
            if len(raw) in (8, 24):
                vals.extend(struct.unpack("<%dd" % (len(raw) // 8), raw))
        elif re.match(rb"F[-+0-9.eE]+$", line):
            vals.append(float(line[1:].decode("ascii")))
    return vals


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    out_dir = Path(sys.argv[2] if len(sys.argv) > 2 else
                   Path(__file__).resolve().parents[1] /
                   "probability-of-collision-for-safe-planning_amd" / "data")
    traj = decode_doubles(ref / "trajectory.dat")
    odom = decode_doubles(ref / "odometry.dat")
    assert len(traj) % 3 == 0 and len(odom) % 3 == 0, (len(traj), len(odom))
    W = len(traj) // 3
    assert len(odom) // 3 == W - 1, (W, len(odom))
    out = out_dir / "pr2test2_plan.txt"
    with open(out, "w") as f:
        f.write("# pocs plan v1: W, then W rows 'x y theta', then W-1 rows 'drot1 dtrans drot2'\n")
        f.write("# decoded (as text, nothing unpickled) from the reference's trajectory.dat / odometry.dat\n")
        f.write("%d\n" % W)
        for i in range(W):
            f.write("%.17g %.17g %.17g\n" % tuple(traj[3 * i:3 * i + 3]))
        for i in range(W - 1):
            f.write("%.17g %.17g %.17g\n" % tuple(odom[3 * i:3 * i + 3]))
    print("wrote", out, "W =", W)


if __name__ == "__main__":
    main()
