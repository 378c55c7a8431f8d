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


# --------------------------------------------------------------------------- #
# golden G8: the reference's own executable VGG-16 (convert_pytorch_vgg.py:13-58)
# --------------------------------------------------------------------------- #
# vgg([...], 3) as convert_pytorch_vgg.py:63 builds it; 'M' = 2x2 s2 pool, 'C' = the same with ceil_mode (identical on even maps)
VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'C', 512, 512, 512, 'M', 512, 512, 512]
# TF names of its 15 convolutions in module order: conv1_1 .. conv5_3, conv6 (3x3 rate 6), conv7 (1x1)
VGG_CONV_NAMES = (['conv%d/conv%d_%d' % (b, b, r) for b, reps in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)) for r in range(1, reps + 1)]
                  + ['conv6', 'conv7'])
# names of the 35 module outputs in module order (a conv's name = its output after the in-place ReLU that follows it)
VGG_TAPS = ['conv1_1', 'conv1_2', 'pool1', 'conv2_1', 'conv2_2', 'pool2', 'conv3_1', 'conv3_2', 'conv3_3', 'pool3',
            'conv4_1', 'conv4_2', 'conv4_3', 'pool4', 'conv5_1', 'conv5_2', 'conv5_3', 'pool5', 'conv6', 'conv7']


def vgg_backbone_weights_oihw(seed):
    """Seeded weights for the reference's torch VGG16 (15 convolutions), in ITS layout: list of (W [O,I,kh,kw], b [O]) fp32.
    He-scaled normal weights (activations stay O(1) through 15 layers), N(0, 0.1) biases; drawn layer by layer from one
    legacy RandomState stream, so make_golden.py and the tests regenerate identical arrays from the seed alone."""
    rs = np.random.RandomState(seed)
    shapes, cin = [], 3
    for v in VGG_CFG:
        if isinstance(v, int):
            shapes.append((v, cin, 3, 3))
            cin = v
    shapes += [(1024, 512, 3, 3), (1024, 1024, 1, 1)]
    out = []
    for (o, i, kh, kw) in shapes:
        w = (rs.randn(o, i, kh, kw) * np.sqrt(2.0 / (i * kh * kw))).astype(np.float32)
        b = (rs.randn(o) * 0.1).astype(np.float32)
        out.append((w, b))
    return out


def vgg_backbone_weights_tf(seed, scope):
    """The same weights as a TF-variable dict: OIHW -> HWIO by np.transpose(w, (2, 3, 1, 0)), the conversion the reference
    itself applies to imported conv weights (nets/caffe_scope.py:57-60)."""
    d = {}
    for name, (w, b) in zip(VGG_CONV_NAMES, vgg_backbone_weights_oihw(seed)):
        d['%s/%s/weights' % (scope, name)] = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))
        d['%s/%s/biases' % (scope, name)] = b
    return d


def vgg_backbone_image(seed, size, batch=1):
    """Seeded NHWC input, the bench's image distribution (uniform 0..255 minus the channel means)."""
    rs = np.random.RandomState(seed)
    img = rs.uniform(0, 255, (batch, size, size, 3)).astype(np.float32)
    return img - np.array([123., 117., 104.], dtype=np.float32)


def g8_sample_index(n):
    """Rows / columns of a map that golden G8 stores: both borders (where SAME padding and the dilated taps matter) and three
    interior positions; G8 also holds float64 sums over each whole tensor."""
    if n <= 8:
        return np.arange(n)
    inner = np.linspace(2, n - 3, 5).round().astype(np.int64)[1:4]
    return np.unique(np.concatenate([[0, 1], inner, [n - 2, n - 1]]))


# golden G8 (second part): the reference's vgg(cfg, i, batch_norm=True) - conv -> BatchNorm2d (inference) -> ReLU
VGG_BN_CFG = [32, 'M', 64]            # two conv + BN + ReLU layers with a pool between them


def vgg_bn_params(seed):
    """Seeded parameters for vgg(VGG_BN_CFG, 3, batch_norm=True): per conv layer (W [O,I,3,3], gamma, beta, running_mean, running_var),
    conv biases zero (slim.conv2d with a normalizer_fn has none, nets/ron_vgg_320.py:595-629)."""
    rs = np.random.RandomState(seed)
    out, cin = [], 3
    for v in VGG_BN_CFG:
        if isinstance(v, int):
            w = (rs.randn(v, cin, 3, 3) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
            gamma = rs.uniform(0.5, 1.5, v).astype(np.float32)
            beta = (rs.randn(v) * 0.1).astype(np.float32)
            mean = (rs.randn(v) * 0.5).astype(np.float32)
            var = rs.uniform(0.5, 1.5, v).astype(np.float32)
            out.append((w, gamma, beta, mean, var))
            cin = v
    return out
