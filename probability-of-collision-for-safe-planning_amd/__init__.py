"""probability-of-collision-for-safe-planning_amd -- MI355X-native collision-probability estimator.

Python host side of libpocs.so (HIP kernels + C ABI, include/pocs.h): a drop-in for the MC and
GMM paths of the reference's OpenRAVE module (mcsimplugin/mcsimplugin.cpp -> MCSimulator.h).
The directory name is not a Python identifier; import it through the alias module `pocs_amd`
at the repo root (or importlib.import_module with the literal name).
"""
from . import planio  # noqa: F401
from .capi import (Context, PocsError, load_library, library_path, OPT_LONE_CALL, OPT_MC_FUSED, OPT_PERSISTENT, OPT_PROFILE, OPT_RUN_AHEAD,  # noqa: F401
                   OPT_STORE_SAMPLES, OPT_USE_GRAPH, OPT_SUB_BATCHES, OPT_MC_NONTEMPORAL, SIGNATURES)
from .planio import DEFAULTS, load_env, load_plan, resample_plan  # noqa: F401

__all__ = ["Context", "PocsError", "load_library", "library_path", "planio", "load_plan", "load_env",
           "resample_plan", "DEFAULTS"]
