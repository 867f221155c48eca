"""Import shim: `import fastnerf` == the package in ./fast-learning-nerf_amd (whose directory
name is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('fast-learning-nerf_amd')
sys.modules[__name__] = _pkg
