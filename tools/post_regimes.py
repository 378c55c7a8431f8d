#!/usr/bin/env python3
"""Post-processing time (ron_post_np, batch 32) across candidate-count regimes: background / objectness biases of
the synthetic head tensors set how many of the 425 k (anchor, class) pairs pass the thresholds.  python tools/post_regimes.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ron_tensorflow_amd import ops, tfe
from ron_tensorflow_amd.nets.ron_vgg_320 import RONNet


def head_tensors(seed, batch, bg, ob, num_classes=21, anchors=10):
    """cls logits N(0,1) with +bg on class 0, objectness logits N(0,1) with +ob on the positive channel, loc N(0,1)."""
    rs = np.random.RandomState(seed)
    cls, obj, loc = [], [], []
    for f in (5, 10, 20, 40):
        c = rs.randn(batch, f, f, anchors, num_classes).astype(np.float32)
        c[..., 0] += bg
        o = rs.randn(batch, f, f, anchors, 2).astype(np.float32)
        o[..., 1] += ob
        cls.append(c); obj.append(o); loc.append(rs.randn(batch, f, f, anchors, 4).astype(np.float32))
    return cls, obj, loc


def clustered_head_tensors(seed, batch, n_clusters, radius, num_classes=21, anchors=10):
    """What a TRAINED detector emits: a few objects, each firing on a neighbourhood of anchors.  Everything is background (class-0 logit
    +12, objectness -8) except `n_clusters` neighbourhoods per image on the 40 x 40 and 20 x 20 maps: (2 * radius + 1)^2 cells x all
    anchors with objectness +8 and ONE class per cluster at logit 14 - 0.4 * distance + noise; loc ~ N(0, 0.05): boxes close to their
    anchors, neighbours overlap heavily (adjacent 64-pixel anchors of the 40 x 40 map: IoU 0.78) -> long class-wise suppression chains."""
    rs = np.random.RandomState(seed)
    cls, obj, loc = [], [], []
    for f in (5, 10, 20, 40):
        c = np.zeros((batch, f, f, anchors, num_classes), np.float32)
        c[..., 0] = 12.0
        o = np.zeros((batch, f, f, anchors, 2), np.float32)
        o[..., 1] = -8.0
        cls.append(c); obj.append(o); loc.append((rs.randn(batch, f, f, anchors, 4) * 0.05).astype(np.float32))
    for b in range(batch):
        for g in range(n_clusters):
            layer = 3 if g % 4 else 2
            f = (5, 10, 20, 40)[layer]
            cy, cx, k = rs.randint(0, f), rs.randint(0, f), rs.randint(1, num_classes)
            for dy in range(-radius, radius + 1):
                for dx in range(-radius, radius + 1):
                    y, x = cy + dy, cx + dx
                    if 0 <= y < f and 0 <= x < f:
                        obj[layer][b, y, x, :, 1] = 8.0
                        cls[layer][b, y, x, :, k] = 14.0 - 0.4 * np.hypot(dy, dx) + rs.randn(anchors) * 0.3
    return cls, obj, loc


def time_post_np(t, adev, reps=10):
    for _ in range(3):
        det, _, ncand = ops.post_np(t[0], t[1], t[2], adev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        det, _, ncand = ops.post_np(t[0], t[1], t[2], adev)
    e1.record()
    torch.cuda.synchronize()
    return det, ncand, e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device('cuda:0')
    adev = ops.anchors_to_device(RONNet().anchors((320, 320)), dev)
    # clustered candidates (round 6): what NMS costs when it actually suppresses - the regimes below keep 381 - 389 of 400 rows
    print('# clustered candidates: n clusters per image x (2 r + 1)^2 cells x 10 anchors, one class per cluster; batch 32')
    for n_clusters, radius in ((5, 4), (5, 2), (20, 2), (20, 1), (80, 1), (80, 0), (1, 6)):
        cls, obj, loc = clustered_head_tensors(11, 32, n_clusters, radius)
        t = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in lst] for lst in (cls, obj, loc)]
        det, ncand, us = time_post_np(t, adev)
        kept = det.count.float()
        print('clusters %3d radius %d: %8.0f candidates / image, %6.1f detections / image (min %d, max %d) of %d sorted rows, %8.1f us per batch of 32'
              % (n_clusters, radius, float(ncand.float().mean()), float(kept.mean()), int(kept.min()), int(kept.max()),
                 int(min(400, float(ncand.float().mean()))), us), flush=True)
    for bg, ob in ((8.0, -4.0), (7.0, -3.0), (6.0, -2.0), (4.0, -2.0), (0.0, 2.0)):
        cls, obj, loc = head_tensors(5, batch=32, bg=bg, ob=ob)
        t = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in lst] for lst in (cls, obj, loc)]
        for _ in range(3):
            det, _, ncand = ops.post_np(t[0], t[1], t[2], adev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            det, _, ncand = ops.post_np(t[0], t[1], t[2], adev)
        e1.record()
        torch.cuda.synchronize()
        us_np = e0.elapsed_time(e1) * 100
        # the TF-evaluation variant on the same heads with eval_ron_network.py's flags (:60-71: select 0.01, nms 0.4, top_k 200, keep 100)
        kw = dict(objectness_thres=0.03, select_threshold=0.01, nms_threshold=0.4, top_k=200, keep_top_k=100,
                  cls_is_prob=False, obj_is_prob=False, loc_decoded=False)
        for _ in range(3):
            sc, bb = tfe.post_tfe(t[0], t[1], t[2], adev, **kw)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            sc, bb = tfe.post_tfe(t[0], t[1], t[2], adev, **kw)
        e1.record()
        torch.cuda.synchronize()
        print('bg %+.0f ob %+.0f: %8.0f candidates / image, %6.1f detections / image, %8.1f us per batch of 32 (np_methods);  '
              'TF variant: %6.1f detections / image, %8.1f us'
              % (bg, ob, float(ncand.float().mean()), float(det.count.float().mean()), us_np,
                 float((sc > 0).sum()) / sc.shape[0], e0.elapsed_time(e1) * 100), flush=True)


if __name__ == '__main__':
    main()
