import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(params=['fp32', 'bf16x3'])
def math_mode(request):
    """Run a GPU test under both matrix-core math modes of the MLP kernels (ops.set_math)."""
    import fastnerf
    old = fastnerf.ops.get_math()
    fastnerf.ops.set_math(request.param)
    yield request.param
    fastnerf.ops.set_math(old)


def noview_state_dicts(golden_dir):
    """The two G19 nets (use_viewdirs=False) as name -> numpy arrays in the reference's state_dict order: trunks from
    g7_weights.npz, `output_linear` from g19_noview.npz, the unused `views_linears.0` zero."""
    import numpy as np
    g7 = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    g19 = np.load(os.path.join(golden_dir, 'g19_noview.npz'))
    out = []
    for pre in ('c.', 'f.'):
        sd = {}
        for n in g19['names.' + pre[0]]:
            n = str(n)
            if n.startswith('pts_linears.'):
                sd[n] = g7[pre + n]
            elif n == 'views_linears.0.weight':
                sd[n] = np.zeros(tuple(g19['shape.' + pre + n]), dtype=np.float32)
            elif n == 'views_linears.0.bias':
                sd[n] = np.zeros(128, dtype=np.float32)
            else:
                sd[n] = g19['w.' + pre + n]
        out.append(sd)
    return out
