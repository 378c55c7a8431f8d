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
    # the fields an N > 1 line carries (round 5): who took part, every rank's own rate, one gather alone, how the weights were made
    assert out['ranks_seen'] == {'world_size': 1, 'distinct_devices': 1, 'device_of_rank': [0], 'image_seed_of_rank': [3]}
    assert out['per_rank_images_per_s']['min'] > 0 and out['per_rank_images_per_s']['max'] >= out['per_rank_images_per_s']['min']
    assert abs(out['per_rank_images_per_s']['max'] - out['value']) / out['value'] < 1e-6       # one rank: its rate is the job's
    assert 0 < out['gather_ms'] < 50
    assert out['weights']['how'].startswith('made') and out['weights']['synthesis_s'] > 0
    # round 6: the sustained leg ran behind the timed region, with the gather in its loop (>= 3 s in windows of 100 steps)
    su = out['sustained']
    assert su['seconds'] >= 2.5 and su['steps'] % su['window_steps'] == 0 and su['steps'] >= 4 * su['window_steps']
    assert len(su['window_images_per_s']['all']) == su['steps'] // su['window_steps'] - 1
    assert 0.5 < su['burst_over_sustained'] < 2.0 and su['images_per_s'] > 0

