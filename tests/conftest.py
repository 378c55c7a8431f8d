import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The C-ABI library is a build product (git-ignored): on a fresh checkout compile it once (hipcc cross-compiles
    gfx950 without a GPU, ~2 min) so that the symbol / host-function tests have something to load."""
    lib = os.path.join(ROOT, 'ron_tensorflow_amd', 'libron_hip.so')
    if not os.path.isfile(lib):
        import subprocess
        subprocess.run(['make', '-C', os.path.join(ROOT, 'ron_tensorflow_amd', 'csrc'), '-j', str(min(8, os.cpu_count() or 1))],
                       check=False)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
