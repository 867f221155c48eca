"""Import shim: `import fastnerf` == the package in ./fast-learning-nerf_amd (whose directory
name is not a valid Python identifier).  Every submodule the package has loaded is registered under the
`fastnerf.` name as well, so that `from fastnerf.render import render` yields the SAME module objects (and
classes: isinstance checks) as `fastnerf.render` -- not a second copy executed under another name."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = 'fast-learning-nerf_amd'
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + '.'):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
