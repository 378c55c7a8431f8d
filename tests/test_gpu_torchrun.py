"""GPU: the multi-process launch path on the hardware that is available to the tests -- one rank.

`bench.py` is started by tests/conftest.py exactly as the driver starts it for N > 1
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ... bench.py --gpus 1`), as a fresh
child process before this test process initialises the GPU.  Under torch.distributed.run the process group (RCCL) exists
even for one rank, so the run goes through the same record packing + `all_gather_into_tensor` + release path as 8 ranks
do (ron_tensorflow_amd/parallel.py); `--check-gather` makes every rank compare the gathered records with its local ones."""
import json

import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_torchrun_gathers_its_own_records(torchrun_child):
    r = torchrun_child
    assert r, 'the torchrun child was not started (no GPU visible at session start?)'
    assert r['rc'] == 0, 'torchrun bench failed (%s):\n%s\n%s' % (' '.join(r['cmd']), r['stdout'][-2000:], r['stderr'][-4000:])
    lines = [l for l in r['stdout'].splitlines() if l.startswith('{') and '"metric"' in l]
    assert len(lines) == 1, r['stdout'][-2000:]
    out = json.loads(lines[0])
    assert out['gather_check'] == 'ok'
    assert out['n_gpus'] == 1 and out['steps'] == 5 and out['scaling'] == 'weak'
    assert out['value'] > 0 and out['config']['mean_detections_per_image'] > 0
