"""Plan / collision-world I/O (text formats written by tools/make_plan_fixture.py).

Reference counterparts: MCSimulation.py:176-198 (loads trajectory.dat / odometry.dat and sends
them by component), gaussprop.py:166-172 (getPathOdometry = inverseOdometry over consecutive
waypoints, MCSimulator.h:434-449) and the obstacle boxes of pr2test2.env.xml.
"""
import math
from pathlib import Path

import numpy as np

DATA = Path(__file__).resolve().parent / "data"

# Parameters of every published run (gaussprop.py:36,39,45-46,56; MCSimulation.py:164,204-205).
DEFAULTS = dict(
    alphas=[0.00025 ** 2, 0.0025 ** 2, 0.0025 ** 2, 0.0025 ** 2],
    Q=0.2 ** 2,
    landmarks=[[3, -3, 0, 0, -3, 3, -3, 3], [0, 0, 2, -2, 2, 2, -2, -2]],
    cov0=[[0.001, 0, 0], [0, 0.001, 0], [0, 0, 0.001]],
    num_particles=10000,
    num_gaussians=3,
)


def load_plan(path=None):
    """Returns dict(traj: W x 3 [x y theta], odom: (W-1) x 3 [drot1 dtrans drot2])."""
    path = Path(path) if path else DATA / "pr2test2_plan.txt"
    rows = [ln.split() for ln in path.read_text().splitlines() if ln.strip() and not ln.startswith("#")]
    W = int(rows[0][0])
    vals = np.array([[float(v) for v in r] for r in rows[1:]], dtype=np.float64)
    if vals.shape != (2 * W - 1, 3):
        raise ValueError("plan file %s: expected %d rows of 3, got %s" % (path, 2 * W - 1, vals.shape))
    return dict(traj=vals[:W].copy(), odom=vals[W:].copy())


def load_env(path=None):
    """Returns dict(footprint: [dx dy hx hy], boxes: M x 5 [cx cy hx hy yaw])."""
    path = Path(path) if path else DATA / "pr2test2_env.txt"
    fp, boxes = [0.0, 0.0, 0.334, 0.334], []
    for ln in path.read_text().splitlines():
        t = ln.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "footprint":
            fp = [float(v) for v in t[1:5]]
        elif t[0] == "box":
            boxes.append([float(v) for v in t[1:6]])
        else:
            raise ValueError("env file %s: unknown record %r" % (path, t[0]))
    return dict(footprint=fp, boxes=np.array(boxes, dtype=np.float64).reshape(-1, 5))


def wrap_angle(a):
    """angleWrap, MCSimulator.h:56-65 (while loops into [0, 2 pi], 2 pi itself kept)."""
    while a < 0:
        a += 2 * math.pi
    while a > 2 * math.pi:
        a -= 2 * math.pi
    return a


def inverse_odometry(p1, p2):
    """inverseOdometry, MCSimulator.h:434-449."""
    r1 = wrap_angle(math.atan2(p2[1] - p1[1], p2[0] - p1[0]) - p1[2])
    tr = math.sqrt((p2[0] - p1[0]) ** 2 + (p2[1] - p1[1]) ** 2)
    r2 = wrap_angle(p2[2] - p1[2] - r1)
    return [r1, tr, r2]


def path_odometry(traj):
    """getPathOdometry, gaussprop.py:166-172: odometry between consecutive waypoints."""
    traj = np.asarray(traj, dtype=np.float64)
    return np.array([inverse_odometry(traj[i], traj[i + 1]) for i in range(len(traj) - 1)])


def resample_plan(plan, W):
    """W-waypoint plan on the same polyline: x, y piecewise linear in the
    index parameter s = j (W0-1)/(W-1); theta piecewise constant from the segment's start
    waypoint; odometry regenerated with inverse_odometry."""
    t0 = np.asarray(plan["traj"], dtype=np.float64)
    W0 = len(t0)
    out = np.zeros((W, 3))
    for j in range(W):
        s = j * (W0 - 1) / (W - 1) if W > 1 else 0.0
        i = min(int(math.floor(s)), W0 - 2) if W0 > 1 else 0
        f = s - i
        if W0 > 1:
            out[j, 0] = t0[i, 0] + f * (t0[i + 1, 0] - t0[i, 0])
            out[j, 1] = t0[i, 1] + f * (t0[i + 1, 1] - t0[i, 1])
            out[j, 2] = t0[i, 2] if f < 1.0 else t0[i + 1, 2]
        else:
            out[j] = t0[0]
    return dict(traj=out, odom=path_odometry(out))
