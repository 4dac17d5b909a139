#!/usr/bin/env python3
"""python tools/run_experiment.py MC|GMM [--runs 200 --particles 10000 ...]  (MCSimulation.py counterpart)"""
import sys
from importlib import import_module
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pocs_amd  # noqa: F401,E402
import_module("probability-of-collision-for-safe-planning_amd.driver").main()
