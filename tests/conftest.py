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
