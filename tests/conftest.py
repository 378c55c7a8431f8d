import os
import sys

import pytest

# the application's job: one hardware queue per stream for the two-slot pipeline tests; read when HIP initialises
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

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


TORCHRUN_CHILD = {}


def pytest_collection_modifyitems(session, config, items):
    """The single-rank torchrun child runs only when tests/test_gpu_torchrun.py is among the SELECTED items (after -m / -k),
    not for every session that mentions gpu."""
    expr = config.getoption('-m') or ''
    if 'gpu' not in expr or 'not gpu' in expr:
        return
    selected = items
    kexpr = config.getoption('-k') or ''
    if kexpr:                                    # -k deselection happens in a later hook: apply the same expression here
        try:
            from _pytest.mark import KeywordMatcher
            from _pytest.mark.expression import Expression
            e = Expression.compile(kexpr)
            selected = [it for it in items if e.evaluate(KeywordMatcher.from_item(it))]
        except Exception:      # noqa: BLE001  (private API moved: fall back to "selected")
            selected = items
    if any(it.nodeid.startswith('tests/test_gpu_torchrun.py') or 'test_gpu_torchrun' in it.nodeid for it in selected):
        _start_torchrun_child()


def _gpu_present():
    """Without torch and without a HIP call (torch.cuda.device_count() may fall back to hipGetDeviceCount, which initialises
    the runtime in THIS process): the kernel driver's device node + a render node."""
    import glob
    return os.path.exists('/dev/kfd') and bool(glob.glob('/dev/dri/renderD*'))


def _start_torchrun_child():
    """`bench.py` under `python -m torch.distributed.run --nproc-per-node 1` as a FRESH child process, started and finished
    before this process touches the GPU (a process that has initialised HIP must not fork + exec on the GPU pool; test modules
    are only imported here, none of them makes a HIP call at import).  tests/test_gpu_torchrun.py reads the result."""
    if TORCHRUN_CHILD or not _gpu_present():
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
