"""CPU, world_size 2 over gloo: the N > 1 path of bench.py (image sharding + one all-gather of records)."""
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
