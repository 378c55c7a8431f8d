"""Oracle: fp32 NHWC restatement of SSD-512 (test infrastructure, see ``oracle/__init__.py``).

Follows ``nets/ssd_vgg_512.py:364-460`` (``ssd_net``), the multibox layer ``nets/ssd_vgg_300.py:403-431``,
``custom_layers.l2_normalization`` / ``pad2d`` (``nets/custom_layers.py:66-163``) and the anchor generator
``nets/ssd_vgg_512.py:286-358``.  TensorFlow graph code: **parity unpinned** for the conv stack (the anchors ARE
pinned: tests/golden/g5_anchors_ssd512.npz comes from the reference's own numpy function).
"""
import math

import numpy as np

from . import np_post
from .ron_forward import F32, conv2d_np, max_pool2x2_np

SCOPE = 'ssd_512_vgg'
FEAT_LAYERS = ['block4', 'block7', 'block8', 'block9', 'block10', 'block11', 'block12']
# SSDNet.default_params, nets/ssd_vgg_512.py:76-102
SSD512 = dict(
    img_shape=(512, 512),
    feat_shapes=[(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)],
    anchor_sizes=[(20.48, 51.2), (51.2, 133.12), (133.12, 215.04), (215.04, 296.96), (296.96, 378.88), (378.88, 460.8),
                  (460.8, 542.72)],
    anchor_ratios=[[2, .5], [2, .5, 3, 1. / 3], [2, .5, 3, 1. / 3], [2, .5, 3, 1. / 3], [2, .5, 3, 1. / 3], [2, .5], [2, .5]],
    anchor_steps=[8, 16, 32, 64, 128, 256, 512],
    anchor_offset=0.5,
    normalizations=[20, -1, -1, -1, -1, -1, -1],
    prior_scaling=[0.1, 0.1, 0.2, 0.2],
)


def anchor_one_layer(img_shape, feat_shape, sizes, ratios, step, offset=0.5, dtype=np.float32):
    """nets/ssd_vgg_512.py:286-338."""
    rows = np.arange(feat_shape[0]).reshape(-1, 1).repeat(feat_shape[1], axis=1)
    cols = np.arange(feat_shape[1]).reshape(1, -1).repeat(feat_shape[0], axis=0)
    y = ((rows.astype(dtype) + offset) * step / img_shape[0])[..., None]
    x = ((cols.astype(dtype) + offset) * step / img_shape[1])[..., None]
    n = len(sizes) + len(ratios)
    h = np.zeros((n,), dtype=dtype)
    w = np.zeros((n,), dtype=dtype)
    h[0] = sizes[0] / img_shape[0]
    w[0] = sizes[0] / img_shape[1]
    di = 1
    if len(sizes) > 1:
        h[1] = math.sqrt(sizes[0] * sizes[1]) / img_shape[0]
        w[1] = math.sqrt(sizes[0] * sizes[1]) / img_shape[1]
        di = 2
    for i, r in enumerate(ratios):
        h[i + di] = sizes[0] / img_shape[0] / math.sqrt(r)
        w[i + di] = sizes[0] / img_shape[1] * math.sqrt(r)
    return y, x, h, w


def anchors_all_layers(p=SSD512):
    return [anchor_one_layer(p['img_shape'], s, p['anchor_sizes'][i], p['anchor_ratios'][i], p['anchor_steps'][i],
                             offset=p['anchor_offset']) for i, s in enumerate(p['feat_shapes'])]


def conv2d_pad_np(x, w, stride=1, rate=1, pad=0):
    """Explicit symmetric zero padding then VALID conv (pad2d + padding='VALID', nets/ssd_vgg_512.py:403-440)."""
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    ho = (h + 2 * pad - ((kh - 1) * rate + 1)) // stride + 1
    wo = (wd + 2 * pad - ((kw - 1) * rate + 1)) // stride + 1
    out = np.zeros((n * ho * wo, cout), dtype=F32)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky * rate: ky * rate + (ho - 1) * stride + 1: stride, kx * rate: kx * rate + (wo - 1) * stride + 1: stride, :]
            out += patch.reshape(-1, cin) @ w[ky, kx]
    return out.reshape(n, ho, wo, cout)


def conv2d_pad_torch(x, w, stride=1, rate=1, pad=0):
    """conv2d_pad_np on the independent torch-CPU operator (cross-check back-end)."""
    import torch
    import torch.nn.functional as Fn
    xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(np.ascontiguousarray(w)).permute(3, 2, 0, 1)
    y = Fn.conv2d(xt, wt, None, stride=stride, padding=pad, dilation=rate)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def max_pool3x3_s1_torch(x):
    import torch
    import torch.nn.functional as Fn
    xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
    return Fn.max_pool2d(xt, 3, 1, padding=1).permute(0, 2, 3, 1).contiguous().numpy()


def max_pool3x3_s1_np(x):
    """slim.max_pool2d [3,3] stride 1 SAME: padding never wins the max."""
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)), constant_values=-np.inf)
    n, h, w, c = x.shape
    out = np.full_like(x, -np.inf)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, xp[:, dy:dy + h, dx:dx + w, :])
    return out


def l2_normalization(x, gamma):
    """nets/custom_layers.py:66-135 with scaling=True: x * rsqrt(max(sum_c x^2, 1e-12)) * gamma."""
    ss = np.sum(x * x, axis=-1, keepdims=True, dtype=F32)
    return (x / np.sqrt(np.maximum(ss, F32(1e-12))) * gamma).astype(F32)


def ssd_forward(images, weights, num_classes=21, round_fn=None, collect=None, backend='numpy', stop_after=None):
    """Returns (predictions, localisations, logits, end_points) like SSDNet.net (nets/ssd_vgg_512.py:459).
    backend 'torch' swaps every conv / pool for the independent torch-CPU operator (tests/test_oracle_forward.py);
    stop_after='block7' returns after the VGG backbone (for the G8 pin: only conv1_1 .. conv7 weights are needed)."""
    from .ron_forward import BACKENDS
    rnd = round_fn if round_fn is not None else (lambda a: a)
    conv_same, _, pool2 = BACKENDS[backend]
    conv_pad = conv2d_pad_np if backend == 'numpy' else conv2d_pad_torch
    pool3 = max_pool3x3_s1_np if backend == 'numpy' else max_pool3x3_s1_torch

    def var(name):
        return np.asarray(weights[SCOPE + '/' + name], dtype=F32)

    def conv(x, scope, stride=1, rate=1, pad=None, relu=True):
        w = var(scope + '/weights')
        if pad is None:
            y = conv_same(rnd(x), rnd(w), stride, rate)
        else:
            y = conv_pad(rnd(x), rnd(w), stride, rate, pad)
        y = y + var(scope + '/biases')
        return np.maximum(y, 0) if relu else y

    end_points = {}
    x = np.asarray(images, dtype=F32)
    # conv1_1 .. conv7 (nets/ssd_vgg_512.py:364-400): PINNED - golden G8 holds every one of these tensors as computed by the
    # reference's own torch VGG16 (convert_pytorch_vgg.py:13-58), tests/test_oracle_forward.py
    for bi, reps in enumerate([2, 2, 3, 3, 3]):
        for r in range(reps):
            x = conv(x, 'conv%d/conv%d_%d' % (bi + 1, bi + 1, r + 1))
            if collect is not None:
                collect['conv%d_%d' % (bi + 1, r + 1)] = x
        end_points['block%d' % (bi + 1)] = x
        x = pool2(x) if bi < 4 else pool3(x)
        if collect is not None:
            collect['pool%d' % (bi + 1)] = x
    x = conv(x, 'conv6', rate=6)
    end_points['block6'] = x
    x = conv(x, 'conv7')
    end_points['block7'] = x
    if collect is not None:
        collect['conv6'], collect['conv7'] = end_points['block6'], end_points['block7']
    if stop_after == 'block7':
        return None, None, None, end_points
    for b in range(8, 13):
        x = conv(x, 'block%d/conv1x1' % b)
        if collect is not None:
            collect['block%d_mid' % b] = x
        if b < 12:
            x = conv(x, 'block%d/conv3x3' % b, stride=2, pad=1)
        else:
            x = conv(x, 'block%d/conv4x4' % b, pad=1)
        end_points['block%d' % b] = x
    predictions, logits, localisations = [], [], []
    for i, layer in enumerate(FEAT_LAYERS):
        net = end_points[layer]
        if SSD512['normalizations'][i] > 0:
            net = l2_normalization(net, var(layer + '_box/L2Normalization/gamma'))
            if collect is not None:
                collect[layer + '_norm'] = net
        a = len(SSD512['anchor_sizes'][i]) + len(SSD512['anchor_ratios'][i])
        loc = conv(net, layer + '_box/conv_loc', relu=False)
        cls = conv(net, layer + '_box/conv_cls', relu=False)
        n, h, w, _ = net.shape
        loc = loc.reshape(n, h, w, a, 4).astype(F32)
        cls = cls.reshape(n, h, w, a, num_classes).astype(F32)
        predictions.append(np_post.softmax_last(cls))
        logits.append(cls)
        localisations.append(loc)
    return predictions, localisations, logits, end_points
