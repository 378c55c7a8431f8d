"""Oracle: fp32 NHWC restatement of the RON-320 conv stack (test infrastructure).

Follows the reference graph code, which is TensorFlow-1.x / tf.contrib.slim and cannot be executed here.  PINNED against the
reference's own executable torch VGG (convert_pytorch_vgg.py:13-58, golden G8, tests/test_oracle_g8.py): conv1_1 .. conv5_3 with
pool1..4, and the conv -> inference BatchNorm -> ReLU layer.  Everything else below (fc6 / fc7, transposed conv, reverse-connection
sum, inception concat + BN, heads) has no executable form in the reference: **parity unpinned** -- restated from the reference
source and cross-checked against the independent torch-CPU operators (tests/test_oracle_forward.py: every layer shape of RON-320
and SSD-512, both back-ends).

  VGG-16 body, fc6/fc7 (both variants)   nets/ron_vgg_320.py:454-483, :530-556
  reverse connection + objectness        nets/ron_vgg_320.py:418-432
  class head (two "inception" blocks)    nets/ron_vgg_320.py:378-404
  box head                               nets/ron_vgg_320.py:406-415
  per-scale loop, softmax, objectness    nets/ron_vgg_320.py:495-506, :568-580
  layer defaults (SAME, ReLU, BN eps)    nets/ron_vgg_320.py:595-629

slim semantics restated here: ``slim.conv2d`` = conv(SAME) -> (+bias | batch_norm) -> ReLU
unless ``activation_fn=None``; a conv with ``normalizer_fn=slim.batch_norm`` has no bias;
inference batch-norm is ``gamma * (x - mean) / sqrt(var + 1e-5) + beta``;
``slim.conv2d_transpose`` kernel 2 stride 2 has bias + ReLU and weights [kh,kw,Cout,Cin];
``slim.max_pool2d`` 2x2 stride 2.  Weights are a dict keyed by TF variable names.

Two back-ends compute the same thing: ``numpy`` (im2col + matmul; the restatement proper) and
``torch`` (F.conv2d etc. on CPU; used as the independent cross-check and as the timed CPU
baseline because it is the faster CPU implementation).
"""
import numpy as np

F32 = np.float32
BN_EPS = 1e-5
SCOPE = 'ron_320_vgg'

VGG_BLOCKS = [('conv1', 2, 64), ('conv2', 2, 128), ('conv3', 3, 256), ('conv4', 3, 512), ('conv5', 3, 512)]
FEAT_LAYERS = ['block7', 'block6', 'block5', 'block4']


# --------------------------------------------------------------------------- #
# primitive ops, numpy back-end (NHWC, fp32)
# --------------------------------------------------------------------------- #
def _same_pad(k, rate):
    total = (k - 1) * rate
    return total // 2, total - total // 2


def conv2d_np(x, w, stride=1, rate=1):
    """SAME conv, x [N,H,W,Cin], w HWIO.  stride > 1 only for kernel == stride on divisible maps."""
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    if stride == 1:
        pt, pb = _same_pad(kh, rate)
        pl, pr = _same_pad(kw, rate)
        xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
        ho, wo = h, wd
    else:
        assert kh == stride and kw == stride and h % stride == 0 and wd % stride == 0 and rate == 1
        xp = x
        ho, wo = h // stride, wd // stride
    out = np.zeros((n * ho * wo, cout), dtype=F32)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky * rate: ky * rate + (ho - 1) * stride + 1: stride,
                       kx * rate: kx * rate + (wo - 1) * stride + 1: stride, :]
            out += patch.reshape(-1, cin) @ w[ky, kx]
    return out.reshape(n, ho, wo, cout)


def conv2d_transpose_np(x, w, stride=2):
    """kernel == stride transposed conv; w [kh,kw,Cout,Cin]: out[2y+ky,2x+kx,co] = sum_ci x[y,x,ci] w[ky,kx,co,ci]."""
    n, h, wd, cin = x.shape
    kh, kw, cout, _ = w.shape
    assert kh == stride and kw == stride
    out = np.zeros((n, h * stride, wd * stride, cout), dtype=F32)
    flat = x.reshape(-1, cin)
    for ky in range(kh):
        for kx in range(kw):
            out[:, ky::stride, kx::stride, :] = (flat @ w[ky, kx].T).reshape(n, h, wd, cout)
    return out


def max_pool2x2_np(x):
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).max(axis=(2, 4))


# --------------------------------------------------------------------------- #
# torch back-end (independent implementation of the same three ops)
# --------------------------------------------------------------------------- #
def _t(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)


def conv2d_torch(x, w, stride=1, rate=1):
    """Same contract as conv2d_np.  TF's SAME rule puts the odd pixel of an even kernel's padding AFTER the data
    (total = (k-1)*rate, before = total // 2), which F.conv2d's symmetric `padding` cannot express: pad explicitly."""
    import torch
    import torch.nn.functional as Fn
    kh, kw = w.shape[:2]
    wt = torch.from_numpy(np.ascontiguousarray(w)).permute(3, 2, 0, 1)
    xt = _t(x)
    if stride == 1:
        pt, pb = _same_pad(kh, rate)
        pl, pr = _same_pad(kw, rate)
        xt = Fn.pad(xt, (pl, pr, pt, pb))
    else:
        assert kh == stride and kw == stride and x.shape[1] % stride == 0 and x.shape[2] % stride == 0 and rate == 1
    y = Fn.conv2d(xt, wt, None, stride=stride, padding=0, dilation=rate)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def conv2d_transpose_torch(x, w, stride=2):
    import torch
    import torch.nn.functional as Fn
    wt = torch.from_numpy(np.ascontiguousarray(w)).permute(3, 2, 0, 1)   # [Cin, Cout, kh, kw]
    y = Fn.conv_transpose2d(_t(x), wt, None, stride=stride)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def max_pool2x2_torch(x):
    import torch.nn.functional as Fn
    return Fn.max_pool2d(_t(x), 2, 2).permute(0, 2, 3, 1).contiguous().numpy()


BACKENDS = {
    'numpy': (conv2d_np, conv2d_transpose_np, max_pool2x2_np),
    'torch': (conv2d_torch, conv2d_transpose_torch, max_pool2x2_torch),
}


# --------------------------------------------------------------------------- #
# slim layer semantics
# --------------------------------------------------------------------------- #
class _Net(object):
    def __init__(self, weights, backend, round_fn=None):
        self.w = weights
        self.conv, self.deconv, self.pool = BACKENDS[backend]
        # optional operand rounding (e.g. to bf16) used to model a reduced-precision device path
        self.rnd = round_fn if round_fn is not None else (lambda a: a)

    def var(self, name):
        return np.asarray(self.w[SCOPE + '/' + name], dtype=F32)

    def bn(self, x, scope):
        g, b = self.var(scope + '/BatchNorm/gamma'), self.var(scope + '/BatchNorm/beta')
        m, v = self.var(scope + '/BatchNorm/moving_mean'), self.var(scope + '/BatchNorm/moving_variance')
        return ((x - m) / np.sqrt(v + F32(BN_EPS)) * g + b).astype(F32)

    def conv_bias(self, x, scope, stride=1, rate=1, relu=True):
        y = self.conv(self.rnd(x), self.rnd(self.var(scope + '/weights')), stride, rate) + self.var(scope + '/biases')
        return np.maximum(y, 0) if relu else y

    def conv_bn_relu(self, x, scope, stride=1):
        y = self.conv(self.rnd(x), self.rnd(self.var(scope + '/weights')), stride, 1)
        return np.maximum(self.bn(y, scope), 0)

    def deconv_bias_relu(self, x, scope):
        y = self.deconv(self.rnd(x), self.rnd(self.var(scope + '/weights')), 2) + self.var(scope + '/biases')
        return np.maximum(y, 0)


def _reverse_module(net, left, right, layer, num_anchors, num_classes, collect=None):
    """nets/ron_vgg_320.py:418-432 for one scale -> (ref_map, objness_logits, cls_logits, loc)."""
    vs = 'reverse_module/%s_reverse' % layer
    if right is None:
        ref = net.conv_bn_relu(left, vs + '_conv_left', stride=2)                 # 2x2 stride 2
    else:
        lc = net.conv_bn_relu(left, vs + '_conv_left')
        up = net.deconv_bias_relu(right, vs + '_deconv_right')
        ref = np.maximum(lc + up, 0)
    obj_h = net.conv_bn_relu(ref, vs + '_objectness')
    keep = {} if collect is None else collect
    keep[layer + '_objectness'] = obj_h
    obj = net.conv_bias(obj_h, vs + '_objectness_score', relu=False)
    # class head: two (3x3 || 1x1) -> concat -> BN -> ReLU blocks, then 3x3 -> A*C   (:378-404)
    x = ref
    for blk in ('_inception1', '_inception2'):
        b0 = net.conv_bias(x, vs + blk + '/Branch_0/Conv2d_3x3', relu=False)
        b1 = net.conv_bias(x, vs + blk + '/Branch_1/Conv2d_1x1', relu=False)
        x = np.maximum(net.bn(np.concatenate([b0, b1], axis=3), vs + blk), 0)
        keep[layer + blk] = x
    cls = net.conv_bias(x, vs + '_inception2/Conv2d_pred_3x3', relu=False)
    # box head (:406-415)
    r = net.conv_bn_relu(ref, vs + '/Conv2d_0_3x3')
    keep[layer + '_reg_hidden'] = r
    loc = net.conv_bias(r, vs + '/Conv2d_1_3x3', relu=False)
    n, h, w, _ = ref.shape
    return (ref, obj.reshape(n, h, w, num_anchors, 2), cls.reshape(n, h, w, num_anchors, num_classes),
            loc.reshape(n, h, w, num_anchors, 4))


def _vgg_body(net, x, end_points, collect=None):
    """conv1_1 .. conv5_3 with the five 2x2 pools (nets/ron_vgg_320.py:454-475) -> pool5.  PINNED: golden G8 holds every one of
    these tensors up to conv5_3 as computed by the reference's own torch VGG16 (tests/test_oracle_forward.py)."""
    for bi, (name, reps, _) in enumerate(VGG_BLOCKS):
        for r in range(reps):
            x = net.conv_bias(x, '%s/%s_%d' % (name, name, r + 1))
            if collect is not None:
                collect['%s_%d' % (name, r + 1)] = x
        end_points['block%d' % (bi + 1)] = x
        x = net.pool(x)
        if collect is not None:
            collect['pool%d' % (bi + 1)] = x
    return x


def vgg_body(images, weights, backend='numpy', round_fn=None):
    """The VGG-16 body alone (what ron_forward runs first), every conv / pool output by name: for the G8 pin."""
    collect = {}
    _vgg_body(_Net(weights, backend, round_fn), np.asarray(images, dtype=F32), {}, collect)
    return collect


def ron_forward(images, weights, variant='reducedfc', num_classes=21, num_anchors=10, backend='numpy',
                round_fn=None, collect=None):
    """RON-320 forward.  images [N,320,320,3] fp32 (mean-subtracted RGB).

    Returns (predictions, logits, objness_pred, objness_logits, localisations, end_points) with the
    list order of the reference: block7 (5x5), block6, block5, block4 (nets/ron_vgg_320.py:580).
    """
    from . import np_post
    net = _Net(weights, backend, round_fn)
    end_points = {}
    x = _vgg_body(net, np.asarray(images, dtype=F32), end_points, collect)
    if variant == 'full':        # nets/ron_vgg_320.py:478-483
        x = net.conv_bias(x, 'fc6')                # 7x7 512 -> 4096
    elif variant == 'reducedfc':  # nets/ron_vgg_320.py:553-556
        x = net.conv_bias(x, 'fc6', rate=3)        # 3x3 rate 3 512 -> 1024
    else:
        raise ValueError('unknown variant %r' % (variant,))
    end_points['block6'] = x
    x = net.conv_bias(x, 'fc7')                    # 1x1
    end_points['block7'] = x

    predictions, logits, objness_pred, objness_logits, localisations = [], [], [], [], []
    ref = None
    for layer in FEAT_LAYERS:
        ref, obj, cls, loc = _reverse_module(net, end_points[layer], ref, layer, num_anchors, num_classes, collect)
        if collect is not None:
            collect[layer + '_ref'] = ref
        predictions.append(np_post.softmax_last(cls))
        logits.append(cls.astype(F32))
        objness_pred.append(np_post.objectness_from_logits(obj))
        objness_logits.append(obj.astype(F32))
        localisations.append(loc.astype(F32))
    return predictions, logits, objness_pred, objness_logits, localisations, end_points


def round_bf16(a):
    """Round-to-nearest-even to bfloat16, returned as float32 (models bf16 operands)."""
    a = np.ascontiguousarray(a, dtype=F32)
    u = a.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    return r.view(F32)


def round_f16(a):
    return np.asarray(a, dtype=np.float16).astype(F32)


def round_f16x3(a):
    """hi + lo with hi = f16(a), lo = f16(a - hi): what a value keeps in the device's split-precision storage
    (RON_DTYPE_F16X3, 22 mantissa bits), returned as float32."""
    a = np.asarray(a, dtype=F32)
    hi = a.astype(np.float16).astype(F32)
    return hi + (a - hi).astype(np.float16).astype(F32)
