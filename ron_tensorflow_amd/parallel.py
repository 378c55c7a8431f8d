"""Image-sharded inference across the GPUs of one node.

The path has no cross-image operation, so a batch is split into contiguous per-rank shards with no
data-path collective; the only exchange is ONE all-gather (RCCL over xGMI; backend "nccl" on ROCm) of the
fixed-size detection records after NMS.  The reference has no inference data parallelism at all
(eval_ron_network.py:93-94 runs batch 1 on one device), so this module has no reference counterpart.

Record layout per image (float32, [top_k + 1, 7]): rows 0..top_k-1 = (class, score, ymin, xmin, ymax, xmax,
anchor_index), zero padded; row top_k = the detection count replicated.  Integers <= 2^24 are exact in fp32.
"""
import torch
import torch.distributed as dist

RECORD_WIDTH = 7


def shard_range(n_images, rank, world_size):
    """Contiguous block of image indices [begin, end) owned by `rank` (remainder spread over the low ranks)."""
    base, rem = divmod(n_images, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def rank_layout(env=None):
    """What a process launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` derives from its environment:
    world size, rank, the GPU it drives (cuda:LOCAL_RANK: one process per GPU of the node) and the seed of ITS synthetic images
    (3 + rank: every rank infers different images, the weights are the same everywhere).  `use_dist`: a process group exists even for one
    rank under torch.distributed.run, so that `--nproc-per-node 1` exercises the gather path."""
    import os
    env = os.environ if env is None else env
    world, rank, local_rank = int(env.get('WORLD_SIZE', '1')), int(env.get('RANK', '0')), int(env.get('LOCAL_RANK', '0'))
    if not (0 <= rank < world and 0 <= local_rank <= rank):
        raise ValueError('inconsistent launch environment: WORLD_SIZE %d RANK %d LOCAL_RANK %d' % (world, rank, local_rank))
    return dict(world=world, rank=rank, local_rank=local_rank, device_index=local_rank, image_seed=3 + rank,
                use_dist=world > 1 or 'RANK' in env)


def shared_host_arrays(make, tag, local_rank, use_dist, directory='/dev/shm'):
    """A dict of numpy arrays every rank of the node needs (bench.py's synthetic weights: 229 M parameters, ~12 s of single-threaded
    RNG): local rank 0 makes it and publishes it as one .npz in `directory` (memory-backed), the others load that after a barrier;
    the file is removed once everyone has it.  Returns (arrays, seconds this rank spent, 'made' | 'loaded')."""
    import os
    import time
    import numpy as np
    t0 = time.perf_counter()
    if not use_dist or dist.get_world_size() == 1:
        arrays = make()
        return arrays, time.perf_counter() - t0, 'made'
    path = os.path.join(directory, 'ron_shared_%s_%s.npz' % (tag, os.environ.get('MASTER_PORT', '0')))
    how = 'loaded'
    if local_rank == 0:
        arrays = make()
        tmp = path + '.tmp.npz'
        np.savez(tmp, **arrays)
        os.replace(tmp, path)
        how = 'made'
    dist.barrier()
    if local_rank != 0:
        with np.load(path) as z:
            arrays = {k: z[k] for k in z.files}
    dist.barrier()
    if local_rank == 0:
        os.remove(path)
    return arrays, time.perf_counter() - t0, how


def pack_records(classes, scores, bboxes, anchor_index, count):
    """Per-image detection lists -> one float32 tensor [N, top_k + 1, 7] (same device)."""
    n, k = scores.shape
    rec = torch.zeros((n, k + 1, RECORD_WIDTH), dtype=torch.float32, device=scores.device)
    rec[:, :k, 0] = classes.to(torch.float32)
    rec[:, :k, 1] = scores
    rec[:, :k, 2:6] = bboxes
    rec[:, :k, 6] = anchor_index.to(torch.float32)
    rec[:, k, :] = count.to(torch.float32)[:, None]
    return rec


def pack_detections(det, out=None):
    """DetectionBuffers -> records [N, top_k + 1, 7] in ONE launch (ron_pack_records); same result as pack_records."""
    import ctypes as C
    from ._lib import check, current_stream, lib, ptr
    if out is None:
        out = torch.empty((det.n, det.capacity + 1, RECORD_WIDTH), dtype=torch.float32, device=det.scores.device)
    d = det.c_struct()
    check(lib().ron_pack_records(C.byref(d), det.n, ptr(out), current_stream()))
    return out


def unpack_records(rec):
    """Inverse of pack_records: (classes int32, scores, bboxes, anchor_index int32, count int32)."""
    k = rec.shape[-2] - 1
    return (rec[..., :k, 0].to(torch.int32), rec[..., :k, 1].contiguous(), rec[..., :k, 2:6].contiguous(),
            rec[..., :k, 6].to(torch.int32), rec[..., k, 0].to(torch.int32))


def gather_detections(rec, group=None, out=None):
    """All-gather equally sized record tensors: [N, K+1, 7] per rank -> [world, N, K+1, 7] on every rank."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
    # the flat "concatenate along dim 0" form is the one every backend accepts (gloo rejects the stacked one)
    dist.all_gather_into_tensor(out.view((world * rec.shape[0],) + tuple(rec.shape[1:])), rec.contiguous(), group=group)
    return out


def bench_loop(pipe, images, steps, warmup, in_flight, detect_args, top_k, rank=0, world=1, use_dist=False, device=None,
               check_gather=True, pack=pack_detections, synchronize=None, consumer_stream=None, before_timed=None,
               after_timed=None, window=0, make_mark=None, mark_ms=None, measure_gather=True, max_ahead=0):
    """The step / consume / gather / check sequence bench.py times, with the pipeline as an argument (bench.py passes a
    DetectPipeline on the GPU; tests/test_parallel_gloo.py a CPU stub at world size 2).

    One step = `pipe.submit(images, **detect_args)`; at most `in_flight` tickets are pending, the oldest is consumed when a
    new one is submitted.  With a process group (`use_dist`) consuming = pack the detections into records on the consumer
    stream, release the slot, all-gather the records into `gathered[world, n, top_k + 1, 7]`.  `warmup` untimed steps,
    then exactly `steps` timed ones bracketed by synchronize + barrier on both sides; the time is the MAX over ranks.
    After the timed region every rank compares the slice of `gathered` that is its own with its local records and checks
    every rank's counts; the verdict is the MIN over ranks, so one bad rank makes it 'MISMATCH' everywhere.

    `window` > 0 (bench.py's sustained leg): a mark goes onto the consuming stream after every `window` timed steps (a timing event
    behind the consumption of the batch that has just been waited for: with F batches in flight the mark of step i stands behind batch
    i - F + 1, so consecutive marks are exactly `window` batches apart), and `window_ms` holds the time between consecutive marks - how
    the rate develops over a run of seconds without the host ever waiting inside it; the first `window` steps lead up to the first mark
    and have no entry (a mark in front of the first submission sits on an idle stream: its timestamp came 0.15 s late when tried).
    `make_mark()` / `mark_ms(a, b)` default to torch timing events (the gloo test passes host clocks).

    `max_ahead` > 0: the host never runs more than that many submitted batches ahead of the GPU (it waits for the completion event of
    batch i - max_ahead before it submits batch i): a serving loop's bounded queue.  0 = unbounded (the host enqueues as fast as it can).

    Returns dict(dt, det (this rank's last detections), gathered, gather_check ('ok' | 'MISMATCH' | None), rank_dt (every rank's own
    time of the timed region), gather_ms (one all-gather of the records alone, after the timed region), window_ms (list, or None))."""
    import contextlib
    import time
    if synchronize is None:
        synchronize = torch.cuda.synchronize
    in_flight = max(1, int(in_flight))
    n = images.shape[0]
    gathered = None
    if use_dist:
        gathered = torch.empty((world, n, top_k + 1, RECORD_WIDTH), dtype=torch.float32, device=device)
    pending, last = [], [None]

    def consumer():
        return torch.cuda.stream(consumer_stream) if consumer_stream is not None else contextlib.nullcontext()

    def consume(ticket):
        if not use_dist:
            last[0] = ticket.wait()                # the current stream waits for that slot; the host does not
            return last[0]
        with consumer():                           # record packing + gather on the consumer's own stream
            det = last[0] = ticket.wait()
            rec = pack(det)
            ticket.release()                       # the slot may overwrite this output set from here on
            gather_detections(rec, out=gathered)
        return det

    submitted = []

    def step():
        if max_ahead > 0 and len(submitted) >= max_ahead:
            done = getattr(submitted.pop(0), '_done', None)
            if done is not None:
                done.synchronize()             # the host waits; the GPU still has max_ahead - 1 batches queued behind this one
        t = pipe.submit(images, **detect_args)
        if max_ahead > 0:
            submitted.append(t)
        pending.append(t)
        return consume(pending.pop(0)) if len(pending) >= in_flight else None

    def drain():
        while pending:
            consume(pending.pop(0))
        return last[0]

    if make_mark is None:
        def make_mark():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            return ev

        def mark_ms(a, b):
            return a.elapsed_time(b)
    marks = []

    def mark():
        with consumer():
            marks.append(make_mark())

    for _ in range(warmup):
        step()
    drain()
    synchronize()
    if use_dist:
        dist.barrier()
    if before_timed is not None:
        before_timed()
    synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        if window > 0 and (i + 1) % window == 0:
            mark()
    det = drain()
    synchronize()
    if use_dist:
        dist.barrier()
    synchronize()
    dt = time.perf_counter() - t0
    if after_timed is not None:
        after_timed(det)
    gather_check = None
    if use_dist and check_gather:
        # the last step's records as this rank packs them vs the slice of the gathered tensor that belongs to this rank
        local = pack(det)
        synchronize()
        same = bool(torch.equal(gathered[rank], local))
        cnt = unpack_records(gathered)[4]
        sane = bool((cnt >= 0).all() and (cnt <= top_k).all() and torch.equal(cnt[rank], det.count.to(cnt.device)))
        flag = torch.tensor([1 if (same and sane) else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_check = 'ok' if int(flag.item()) == 1 else 'MISMATCH'
    window_ms = [mark_ms(a, b) for a, b in zip(marks[:-1], marks[1:])] if window > 0 else None
    rank_dt, gather_ms = [dt], None
    if use_dist and not measure_gather:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    elif use_dist:
        # every rank's own time (the reported one is their MAX) and what ONE gather of the records costs on its own
        mine = torch.tensor([dt], dtype=torch.float64, device=device)
        every = torch.empty((world,), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every, mine)
        rank_dt = [float(x) for x in every.tolist()]
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rec = pack(det)
        scratch = torch.empty_like(gathered)       # (not `gathered`: that holds what the timed region's last step gathered)
        synchronize()
        dist.barrier()
        g0 = time.perf_counter()
        for _ in range(5):
            with consumer():
                gather_detections(rec, out=scratch)
        synchronize()
        gather_ms = (time.perf_counter() - g0) / 5 * 1e3
    return dict(dt=dt, det=det, gathered=gathered, gather_check=gather_check, rank_dt=rank_dt, gather_ms=gather_ms, window_ms=window_ms)
