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
    _start_torchrun_child(session.config)


TORCHRUN_CHILD = {}


def _gpu_run_selected(config):
    expr = config.getoption('-m') or ''
    return 'gpu' in expr and 'not gpu' not in expr


def _start_torchrun_child(config):
    """`bench.py` under `python -m torch.distributed.run --nproc-per-node 1` as a FRESH child process, started and finished
    before this process touches the GPU (a process that has initialised HIP must not fork + exec on the GPU pool; counting
    devices does not initialise it).  tests/test_gpu_torchrun.py reads the result."""
    if not _gpu_run_selected(config) or TORCHRUN_CHILD:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:        # a free rendezvous port on the loopback interface
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2',
           '--no-cpu-baseline', '--check-gather']
    try:
        p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        TORCHRUN_CHILD.update(rc=p.returncode, stdout=p.stdout.decode(errors='replace'), stderr=p.stderr.decode(errors='replace'), cmd=cmd)
    except Exception as e:      # noqa: BLE001  (reported by the test)
        TORCHRUN_CHILD.update(rc=-1, stdout='', stderr=repr(e), cmd=cmd)


@pytest.fixture(scope='session')
def torchrun_child():
    return TORCHRUN_CHILD


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
