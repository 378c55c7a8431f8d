"""Oracle: RON anchor grids (test infrastructure, see ``oracle/__init__.py``).

Restates reference ``nets/ron_vgg_320.py:285-333`` (one layer) and ``:336-355``
(all layers).  The arithmetic order and dtypes are kept because the grids are
float32 and the decode is compared to 1e-4:

* centres:  ``((idx.astype(f32) + offset) * step) / img``   -- float32 all the way
* sizes:    ``s / img / sqrt(r)`` and ``s / img * sqrt(r)`` -- python doubles, cast
  to float32 on store; anchor index ``a = i_ratio * len(sizes) + j_size``.
"""
import math

import numpy as np

# RONNet.default_params, reference nets/ron_vgg_320.py:97-124
RON320 = dict(
    img_shape=(320, 320),
    feat_shapes=[(5, 5), (10, 10), (20, 20), (40, 40)],
    anchor_sizes=[(224., 256.), (160., 192.), (96., 128.), (32., 64.)],
    anchor_ratios=[[1, 2, 3, 1. / 2, 1. / 3]] * 4,
    anchor_steps=[64, 32, 16, 8],
    anchor_offset=0.5,
    prior_scaling=[0.1, 0.1, 0.2, 0.2],
)


def anchor_one_layer(img_shape, feat_shape, sizes, ratios, step, offset=0.5,
                     dtype=np.float32):
    """(y, x, h, w) for one feature map; y, x: [H, W, 1], h, w: [A].

    Reference: nets/ron_vgg_320.py:285-333.
    """
    rows = np.arange(feat_shape[0]).reshape(-1, 1).repeat(feat_shape[1], axis=1)
    cols = np.arange(feat_shape[1]).reshape(1, -1).repeat(feat_shape[0], axis=0)
    y = ((rows.astype(dtype) + offset) * step) / img_shape[0]
    x = ((cols.astype(dtype) + offset) * step) / img_shape[1]
    y = y[..., None]
    x = x[..., None]
    n_sizes = len(sizes)
    h = np.zeros((n_sizes * len(ratios),), dtype=dtype)
    w = np.zeros_like(h)
    for i_ratio, r in enumerate(ratios):
        root = math.sqrt(r)
        for j_size, s in enumerate(sizes):
            a = i_ratio * n_sizes + j_size
            h[a] = s / img_shape[0] / root
            w[a] = s / img_shape[1] * root
    return y, x, h, w


def anchors_all_layers(img_shape=RON320['img_shape'],
                       feat_shapes=RON320['feat_shapes'],
                       anchor_sizes=RON320['anchor_sizes'],
                       anchor_ratios=RON320['anchor_ratios'],
                       anchor_steps=RON320['anchor_steps'],
                       offset=RON320['anchor_offset'],
                       dtype=np.float32):
    """List of per-layer (y, x, h, w).  Reference: nets/ron_vgg_320.py:336-355."""
    return [anchor_one_layer(img_shape, fs, anchor_sizes[i], anchor_ratios[i],
                             anchor_steps[i], offset=offset, dtype=dtype)
            for i, fs in enumerate(feat_shapes)]
