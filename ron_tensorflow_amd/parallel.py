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
