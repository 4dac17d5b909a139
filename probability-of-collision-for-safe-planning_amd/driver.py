"""Experiment driver and report writer (row N2 of SURVEY 8f): the Python-3 counterpart of
MCSimulation.py:100-270 on top of the text channel of libpocs.so.

Same command sequence (MCSimulation.py:154-207), same run loop with a per-run journal that is
flushed and fsync'ed (:226-256), same report layout (writeReport / writeReportGMM, :16-77) so the
output can be diffed against finalpaper/analysis/*Report*.txt.  What differs, on purpose: the
collision world is the explicit 2-D model (an env file instead of OpenRAVE), every run gets a
reproducible seed, and times are wall-clock seconds (the reference's time.clock() is process
CPU time on Linux/Python 2).
"""
import datetime
import os
import time
from pathlib import Path

import numpy as np

from . import planio
from .capi import Context


def list2string(values):
    """list2String, MCSimulation.py:81-85."""
    return "".join(str(v) + " " for v in values)


def push_configuration(mod, plan, env, params, num_particles, simoption, num_gaussians):
    """MCSimulation.py:154-207, command for command."""
    mod.SendCommand("clearObstacles")            # the world description starts from an explicit (empty) table
    for b in np.asarray(env["boxes"]).reshape(-1, 5):
        mod.SendCommand("addObstacle " + list2string(repr(float(v)) for v in b))
    mod.SendCommand("setFootprint " + list2string(repr(float(v)) for v in env["footprint"]))
    mod.SendCommand("setAlphas " + list2string(repr(float(v)) for v in params["alphas"]))
    mod.SendCommand("setQ " + repr(float(params["Q"])))
    lm = np.asarray(params["landmarks"], dtype=np.float64)
    mod.SendCommand("setNumLandmarks " + str(lm.shape[1]))
    mod.SendCommand("setLandmarks " + list2string(repr(float(v)) for v in lm[0]) +
                    list2string(repr(float(v)) for v in lm[1]))
    mod.SendCommand("setNumParticles " + str(num_particles))
    cov = np.asarray(params["cov0"], dtype=np.float64)
    mod.SendCommand("setInitialCovariance " + "".join(list2string(repr(float(v)) for v in r) for r in cov))
    traj = np.asarray(plan["traj"], dtype=np.float64).T
    odom = np.asarray(plan["odom"], dtype=np.float64).T
    mod.SendCommand("setPathLength " + str(traj.shape[1]))
    mod.SendCommand("setTrajectory " + "".join(list2string(repr(float(v)) for v in r) for r in traj))
    mod.SendCommand("setOdometry " + "".join(list2string(repr(float(v)) for v in r) for r in odom))
    if simoption == "GMM":
        mod.SendCommand("setNumGaussians " + str(num_gaussians))
        mod.SendCommand("setNumGMMSamples " + str(num_particles))


def write_report(path, simoption, envfile, params, num_runs, num_particles, plan, times, props,
                 num_gaussians=None):
    """writeReport / writeReportGMM field layout, MCSimulation.py:16-77."""
    lm = np.asarray(params["landmarks"])
    with open(path, "w") as f:
        f.write("Environment: " + str(envfile) + "\n")
        f.write("Num Landmarks: " + str(lm.shape[1]) + "\n")
        f.write("Landmarks: \n" + str(lm) + "\n")
        f.write("Alphas: \n" + list2string(params["alphas"]) + "\n")
        f.write("Sensor Noise Variance: " + str(params["Q"]) + "\n")
        f.write("Initial Covariance: \n" + str(np.asarray(params["cov0"])) + "\n")
        f.write("---------------------------------\n")
        f.write("NumSimulations: " + str(num_runs) + "\n")
        if simoption == "MC":
            f.write("Num Particles: " + str(num_particles) + "\n")
        else:
            f.write("Num Samples: " + str(num_particles) + "\n")
            f.write("Num Gaussians: " + str(num_gaussians) + "\n")
        f.write("Simulation Times: \n" + str(list(times)) + "\n")
        f.write("Collision Proportions: \n" + str(list(props)) + "\n")
        f.write("Average Sim Time: " + str(float(np.average(times))) + "\n")
        f.write("Average Prob Collision: " + str(float(np.average(props))) + "\n")
        f.write("---------------------------------\n")
        f.write("Trajectory: \n" + str(np.asarray(plan["traj"])) + "\n")
        f.write("Odometry: \n" + str(np.asarray(plan["odom"])) + "\n")


def summary(props, times):
    """Mean / sample standard deviation / range, as finalpaper/analysis/plotData.m:10-42 reports."""
    p, t = np.asarray(props, dtype=np.float64), np.asarray(times, dtype=np.float64)
    return dict(mean=float(p.mean()), std=float(p.std(ddof=1)) if len(p) > 1 else 0.0,
                min=float(p.min()), max=float(p.max()), mean_time=float(t.mean()))


def run_experiment(simoption, num_runs=200, num_particles=10000, num_gaussians=3, seed=1, device=0,
                   plan=None, env=None, envfile=None, params=None, out_dir=".", stamp=None, batch=1, run_ahead=0):
    """MCSimulation.py:221-269.  Returns dict(times, proportions, journal, report, summary).

    batch=1 issues one run per command like the reference.  batch=R (ours) advances R of the
    independent runs per command in lockstep on the GPU (pocs_set_batch): same runs, same seeds,
    same journal and report; a run's simTime is then its call's wall time / R.
    run_ahead keeps one command per run -- the reference's loop, unchanged -- but lets the library evaluate the next R
    runs in one launch and answer the following commands from it (setRunAhead): 0 (the default, as in the OpenRAVE
    adapter) = R sized by the library from the sample count (64 at the reference's own 10^4 samples), 1 = off, one
    launch per run.  Same runs, same seeds, same journal, bit for bit (tests/test_driver.py); a command's simTime is
    what that command took -- the first of a group does the group's work, the others return at once."""
    if simoption not in ("MC", "GMM"):
        raise ValueError('simoption must be "MC" or "GMM"')              # MCSimulation.py:108-111
    plan = plan or planio.load_plan()
    envfile = envfile or str(planio.DATA / "pr2test2_env.txt")
    env = env or planio.load_env(envfile)
    params = params or planio.DEFAULTS
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    st = stamp or datetime.datetime.now().strftime("%Y-%m-%d_%H_%M_%S")
    journal = out_dir / (("checkpoint_" if simoption == "MC" else "GMMcheckpoint_") + st + ".txt")
    report = out_dir / (("simReport_" if simoption == "MC" else "GMMsimReport_") + st + ".txt")
    times, props = [], []
    with Context(device) as mod, open(journal, "w") as f2:
        push_configuration(mod, plan, env, params, num_particles, simoption, num_gaussians)
        mod.SendCommand("setSeed " + str(int(seed)))
        if int(run_ahead) != 1:                      # 0 = sized by the library from the sample count
            mod.SendCommand("setRunAhead " + str(int(run_ahead)))
        command = "runSimulation" if simoption == "MC" else "runGMMEstimation"
        batch = max(1, int(batch))
        i = 0

        def journal_runs(entries):
            nonlocal i
            for sim_time, p in entries:
                times.append(sim_time)
                props.append(p)
                f2.write("Simulation: " + str(i) + "\n")
                f2.write("simTime: " + str(sim_time) + "\n")
                f2.write("collProp: " + str(p) + "\n")
                i += 1
            f2.flush()
            os.fsync(f2.fileno())

        while i < num_runs:
            b = min(batch, num_runs - i)
            if batch > 1:
                mod.set_batch(b)
            start = time.perf_counter()
            collprop = float(mod.SendCommand(command))
            wall = time.perf_counter() - start
            # (run-ahead: a command's simTime is what THAT command took, as the reference's loop measures it -- the command
            # that launched a group carries the group's work, the ones answered from it take microseconds; their mean, the
            # report's "Average Sim Time", is the experiment's time per run either way)
            journal_runs([(wall / b, float(v)) for v in mod.batch_probabilities()] if batch > 1 else [(wall, collprop)])
    write_report(report, simoption, envfile, params, num_runs, num_particles, plan, times, props, num_gaussians)
    return dict(times=times, proportions=props, journal=journal, report=report, summary=summary(props, times))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="MC / GMM collision-probability experiment (MCSimulation.py counterpart)")
    ap.add_argument("simoption", choices=["MC", "GMM"])
    ap.add_argument("--runs", type=int, default=200)
    ap.add_argument("--particles", type=int, default=10000)
    ap.add_argument("--gaussians", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1, help="runs advanced in lockstep per command (1 = like the reference)")
    ap.add_argument("--run-ahead", type=int, default=0, help="one command per run, the next R runs evaluated in one launch (0 = sized by the library, 1 = off)")
    ap.add_argument("--plan", default=None)
    ap.add_argument("--env", default=None)
    ap.add_argument("--out", default=".")
    a = ap.parse_args(argv)
    r = run_experiment(a.simoption, a.runs, a.particles, a.gaussians, a.seed, a.device,
                       plan=planio.load_plan(a.plan) if a.plan else None, envfile=a.env, out_dir=a.out, batch=a.batch, run_ahead=a.run_ahead)
    s = r["summary"]
    print("Average Prob Collision: %.6f  (sd %.6f, range %.4f-%.4f), Average Sim Time: %.6f s"
          % (s["mean"], s["std"], s["min"], s["max"], s["mean_time"]))
    print("journal:", r["journal"], " report:", r["report"])


if __name__ == "__main__":
    main()
