"""Import alias: `import pocs_amd` loads the package directory
`probability-of-collision-for-safe-planning_amd/` (whose name is not a Python identifier)."""
import importlib
import sys
from pathlib import Path

_root = str(Path(__file__).resolve().parent)
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("probability-of-collision-for-safe-planning_amd")
sys.modules[__name__] = _pkg
