"""Seeded synthetic head tensors for parity tests (test infrastructure).

Legacy ``np.random.RandomState`` streams are stable across numpy versions, so
fixtures store only the seed and the generator parameters.
"""
import numpy as np

from . import anchors as _anchors

RON_FEAT_SHAPES = _anchors.RON320['feat_shapes']


def head_tensors(seed, batch=1, feat_shapes=RON_FEAT_SHAPES, num_anchors=10,
                 num_classes=21, bg=8.0, ob=-4.0, cls_scale=1.0):
    """Per-layer (cls_logits [B,H,W,A,C], objness_logits [B,H,W,A,2], loc [B,H,W,A,4]).

    cls logits ~ N(0, cls_scale) with ``+bg`` on class 0, objectness logits ~ N(0,1)
    with ``+ob`` on the positive channel, loc ~ N(0,1).  (bg, ob) = (+8, -4) gives
    about a thousand candidates per image above 0.01, (+4, -2) about 190 k.
    """
    rs = np.random.RandomState(seed)
    cls_l, obj_l, loc_l = [], [], []
    for (h, w) in feat_shapes:
        cls = (rs.randn(batch, h, w, num_anchors, num_classes) * cls_scale).astype(np.float32)
        cls[..., 0] += np.float32(bg)
        obj = rs.randn(batch, h, w, num_anchors, 2).astype(np.float32)
        obj[..., 1] += np.float32(ob)
        loc = rs.randn(batch, h, w, num_anchors, 4).astype(np.float32)
        cls_l.append(cls)
        obj_l.append(obj)
        loc_l.append(loc)
    return cls_l, obj_l, loc_l


SSD512_FEAT_SHAPES = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
SSD512_ANCHORS = [4, 6, 6, 6, 6, 4, 4]


def ssd_head_tensors(seed, batch=1, num_classes=21, bg=6.0, cls_scale=1.0):
    """SSD-512 head tensors (7 scales, 4 / 6 anchors per cell = 24 564 anchors, no objectness):
    per-layer (cls_logits [B,H,W,A,C], loc [B,H,W,A,4])."""
    rs = np.random.RandomState(seed)
    cls_l, loc_l = [], []
    for (h, w), a in zip(SSD512_FEAT_SHAPES, SSD512_ANCHORS):
        cls = (rs.randn(batch, h, w, a, num_classes) * cls_scale).astype(np.float32)
        cls[..., 0] += np.float32(bg)
        cls_l.append(cls)
        loc_l.append(rs.randn(batch, h, w, a, 4).astype(np.float32))
    return cls_l, loc_l
