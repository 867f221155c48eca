"""fast-learning-nerf_amd -- MI355X-native NeRF training inner loop.

Host-side mirror of the reference's renderer / quadtree surface
(nerf-ours/{run_nerf,render,run_nerf_helpers,model,tree}.py) on top of
libfastnerf.so (hand-written HIP for gfx950, C ABI in include/fastnerf.h).
The directory name contains '-', import it with
    importlib.import_module('fast-learning-nerf_amd')
or through the `fastnerf` shim module at the repository root.
"""
from . import _lib, ops  # noqa: F401
from . import run_nerf_helpers, model, render, run_nerf, tree, parallel, synthetic, nerfpp, torch_ops  # noqa: F401
from .build import build  # noqa: F401

__all__ = ['ops', 'run_nerf_helpers', 'model', 'render', 'run_nerf', 'tree', 'parallel', 'build']
