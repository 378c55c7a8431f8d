"""CPU, world_size 2 and 8 over gloo: the N > 1 path of bench.py (rank layout, shared weights, image sharding, one all-gather of
records, the timed loop with its barriers and MAX-over-ranks time)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ron_tensorflow_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_detections(rank, n, k):
    g = torch.Generator().manual_seed(100 + rank)
    count = torch.randint(0, k + 1, (n,), generator=g, dtype=torch.int32)
    classes = torch.randint(1, 21, (n, k), generator=g, dtype=torch.int32)
    scores = torch.rand((n, k), generator=g)
    boxes = torch.rand((n, k, 4), generator=g)
    anchor = torch.randint(0, 21250, (n, k), generator=g, dtype=torch.int32)
    mask = torch.arange(k)[None, :] < count[:, None]
    return classes * mask, scores * mask, boxes * mask[..., None], anchor * mask, count


def _worker(rank, world, port, n, k, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        det = _fake_detections(rank, n, k)
        rec = parallel.pack_records(*det)
        allrec = parallel.gather_detections(rec)
        assert allrec.shape == (world, n, k + 1, parallel.RECORD_WIDTH)
        ok = True
        for r in range(world):
            want = _fake_detections(r, n, k)
            got = parallel.unpack_records(allrec[r])
            for a, b in zip(got, want):
                ok = ok and torch.equal(a.to(b.dtype), b)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_gather_detections_world2():
    world, n, k = 2, 3, 400
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, k, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ---- bench.py's timed loop (parallel.bench_loop) at world size 2, with a stub pipeline ------------------------------
class _StubTicket(object):
    def __init__(self, det, log):
        self.det, self.log = det, log

    def wait(self):
        return self.det

    def release(self):
        self.log.append('release')


class _StubPipeline(object):
    """submit() -> a ticket whose detections depend on (rank, step); rank r sleeps r * 20 ms per step (ranks finish apart)."""

    def __init__(self, rank, n, k, sleep_s=0.02):
        self.rank, self.n, self.k, self.step, self.log, self.sleep_s = rank, n, k, 0, [], sleep_s

    def submit(self, images, **detect_args):
        import time
        from ron_tensorflow_amd.ops import DetectionBuffers
        assert detect_args == dict(select_threshold=0.01, nms_threshold=0.45)
        time.sleep(self.sleep_s * self.rank)
        det = DetectionBuffers(self.n, self.k, 'cpu')
        cl, sc, bb, ai, cnt = _fake_detections(1000 * self.rank + self.step, self.n, self.k)
        det.classes, det.scores, det.bboxes, det.anchor_index, det.count = cl.to(torch.int32), sc, bb, ai.to(torch.int32), cnt
        self.step += 1
        return _StubTicket(det, self.log)


def _pack_cpu(det):
    return parallel.pack_records(det.classes, det.scores, det.bboxes, det.anchor_index, det.count)


def _bench_worker(rank, world, port, corrupt_rank, ret, sleep_s=0.02):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n, k, steps, warmup, in_flight = 3, 400, 4, 2, 2
        pipe = _StubPipeline(rank, n, k, sleep_s)
        images = torch.zeros((n, 8, 8, 3))

        def after_timed(det):
            if rank == corrupt_rank:                  # this rank's local records no longer equal what it sent
                det.scores[0, 0] += 1.0

        res = parallel.bench_loop(pipe, images, steps, warmup, in_flight, dict(select_threshold=0.01, nms_threshold=0.45), k,
                                  rank=rank, world=world, use_dist=True, device='cpu', pack=_pack_cpu, synchronize=lambda: None,
                                  after_timed=after_timed)
        assert pipe.step == warmup + steps and pipe.log.count('release') == warmup + steps
        # every rank holds every rank's records of the LAST step
        found = []
        for r in range(world):
            want = parallel.pack_records(*_fake_detections(1000 * r + warmup + steps - 1, n, k))
            found.append(bool(torch.equal(res['gathered'][r], want)))
        # bench.py's sustained leg: the same loop without warm-up, a mark every `window` steps, no gather check / gather timing
        import time
        sus = parallel.bench_loop(pipe, images, 6, 0, in_flight, dict(select_threshold=0.01, nms_threshold=0.45), k,
                                  rank=rank, world=world, use_dist=True, device='cpu', pack=_pack_cpu, synchronize=lambda: None,
                                  check_gather=False, window=2, make_mark=time.perf_counter, mark_ms=lambda a, b: (b - a) * 1e3,
                                  measure_gather=False)
        ret[rank] = dict(dt=res['dt'], gather_check=res['gather_check'], found=found, rank_dt=res['rank_dt'], gather_ms=res['gather_ms'],
                         sustained=dict(dt=sus['dt'], window_ms=sus['window_ms'], gather_check=sus['gather_check'], gather_ms=sus['gather_ms'],
                                        steps=pipe.step - warmup - steps),
                         det_is_last=bool(torch.equal(res['det'].count, _fake_detections(1000 * rank + warmup + steps - 1, n, k)[4])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('corrupt_rank', [-1, 1])
def test_bench_loop_world2(corrupt_rank):
    """The N > 1 control flow of bench.py without a node: rank-indexed `gathered[rank]`, the MAX all-reduce of the time, the
    barriers inside the timed region, the MIN all-reduce of the gather check."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bench_worker, args=(world, _free_port(), corrupt_rank, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0['found'] == [True, True] and r1['found'] == [True, True]        # own and the other rank's records
    assert r0['det_is_last'] and r1['det_is_last']
    assert r0['dt'] == r1['dt']                                               # MAX over ranks, the same on both
    assert r0['dt'] >= 4 * 0.02                                               # ... and it is the slow rank's (4 steps x 20 ms)
    want = 'ok' if corrupt_rank < 0 else 'MISMATCH'
    assert r0['gather_check'] == want and r1['gather_check'] == want          # one bad rank fails the check on every rank
    # the sustained leg: 6 steps in windows of 2 -> 3 marks, 2 window times, within the leg's time; MAX over ranks; nothing else
    for r in (r0, r1):
        s = r['sustained']
        assert s['steps'] == 6 and len(s['window_ms']) == 2 and all(w >= 0 for w in s['window_ms'])
        assert s['gather_check'] is None and s['gather_ms'] is None
        assert sum(s['window_ms']) * 1e-3 <= s['dt'] + 1e-3
    assert r0['sustained']['dt'] == r1['sustained']['dt'] and r1['sustained']['dt'] >= 6 * 0.02
    assert sum(r1['sustained']['window_ms']) >= 3 * 20 - 1                    # rank 1 sleeps 20 ms per submission


def test_bench_loop_world8():
    """The launch the driver makes on an 8-GPU node (`--nproc-per-node 8`), rehearsed on CPU: eight ranks through the whole timed loop.
    Every rank ends up with every rank's records, the time is the slowest rank's on all of them, the per-rank times and the cost of one
    gather are reported."""
    world, sleep_s = 8, 0.005
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bench_worker, args=(world, _free_port(), -1, ret, sleep_s), nprocs=world, join=True)
    assert sorted(ret.keys()) == list(range(world))
    for r in range(world):
        assert ret[r]['found'] == [True] * world and ret[r]['det_is_last'] and ret[r]['gather_check'] == 'ok'
        assert ret[r]['dt'] == ret[0]['dt']                                   # MAX over ranks, identical everywhere
        assert len(ret[r]['rank_dt']) == world and ret[r]['rank_dt'] == ret[0]['rank_dt']
        assert max(ret[r]['rank_dt']) == ret[r]['dt'] and min(ret[r]['rank_dt']) > 0
        assert ret[r]['gather_ms'] is not None and ret[r]['gather_ms'] > 0
    assert ret[0]['dt'] >= 4 * sleep_s * (world - 1)                          # rank 7 sleeps 7 x 5 ms in each of the 4 timed steps


def test_rank_layout_of_an_eight_gpu_node():
    """RANK / LOCAL_RANK / WORLD_SIZE as torch.distributed.run sets them -> one GPU and one image seed per rank."""
    lays = [parallel.rank_layout({'WORLD_SIZE': '8', 'RANK': str(r), 'LOCAL_RANK': str(r), 'MASTER_ADDR': '127.0.0.1'}) for r in range(8)]
    assert [l['device_index'] for l in lays] == list(range(8))                # cuda:LOCAL_RANK, every GPU exactly once
    assert [l['image_seed'] for l in lays] == [3 + r for r in range(8)]       # different images on every rank
    assert all(l['use_dist'] and l['world'] == 8 for l in lays)
    one = parallel.rank_layout({})                                            # plain `python bench.py`
    assert one == dict(world=1, rank=0, local_rank=0, device_index=0, image_seed=3, use_dist=False)
    assert parallel.rank_layout({'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})['use_dist']     # --nproc-per-node 1: the gather path
    with pytest.raises(ValueError):
        parallel.rank_layout({'WORLD_SIZE': '2', 'RANK': '2', 'LOCAL_RANK': '0'})


def _share_worker(rank, world, port, tmpdir, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        calls = []

        def make():
            calls.append(1)
            g = np.random.RandomState(7)
            return {'ron_320_vgg/conv1/conv1_1/weights': g.rand(3, 3, 3, 64).astype(np.float32), 'b': g.rand(5).astype(np.float32)}

        arrays, secs, how = parallel.shared_host_arrays(make, 'test', rank, True, directory=tmpdir)
        ret[rank] = dict(made=len(calls), how=how, sum=float(sum(v.astype(np.float64).sum() for v in arrays.values())),
                         keys=sorted(arrays), left=sorted(os.listdir(tmpdir)))
    finally:
        dist.destroy_process_group()


def test_shared_host_arrays_made_once_per_node(tmp_path):
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_share_worker, args=(world, _free_port(), str(tmp_path), ret), nprocs=world, join=True)
    assert [ret[r]['made'] for r in range(world)] == [1, 0, 0] and [ret[r]['how'] for r in range(world)] == ['made', 'loaded', 'loaded']
    assert ret[0]['sum'] == ret[1]['sum'] == ret[2]['sum'] and ret[0]['keys'] == ret[1]['keys'] == ['b', 'ron_320_vgg/conv1/conv1_1/weights']
    assert os.listdir(str(tmp_path)) == []                                    # the file is gone once everyone has the arrays


def test_shard_range_partitions_every_image_once():
    for n in (1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, e = parallel.shard_range(n, r, world)
                assert 0 <= b <= e <= n
                seen.extend(range(b, e))
            assert seen == list(range(n))
            sizes = [parallel.shard_range(n, r, world)[1] - parallel.shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip_exact_integers():
    det = _fake_detections(5, 4, 400)
    got = parallel.unpack_records(parallel.pack_records(*det))
    for a, b in zip(got, det):
        assert torch.equal(a.to(b.dtype), b)
    assert float(np.float32(21249)) == 21249.0
