"""Weight contract of the RON-320 graph: TF variable names/shapes, synthetic initialisation, .npz IO.

The reference ships no checkpoint (README.md:23,28 are Google-Drive links), so benchmarks and
tests use seeded synthetic weights of the exact architecture.  Names follow the TF variable
scopes of nets/ron_vgg_320.py:378-580.
"""
import numpy as np

SCOPE = 'ron_320_vgg'
FEAT_LAYERS = ['block7', 'block6', 'block5', 'block4']


def variable_shapes(variant='reducedfc', num_classes=21, num_anchors=10):
    """Ordered dict-like list of (tf_name, shape) the graph expects."""
    c6 = {'full': 4096, 'reducedfc': 1024}[variant]
    out = []

    def conv(scope, k, cin, cout, bias=True):
        out.append(('%s/%s/weights' % (SCOPE, scope), (k, k, cin, cout)))
        if bias:
            out.append(('%s/%s/biases' % (SCOPE, scope), (cout,)))

    def bn(scope, ch):
        for n in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
            out.append(('%s/%s/BatchNorm/%s' % (SCOPE, scope, n), (ch,)))

    cin = 3
    for b, (reps, width) in enumerate([(2, 64), (2, 128), (3, 256), (3, 512), (3, 512)]):
        for r in range(reps):
            conv('conv%d/conv%d_%d' % (b + 1, b + 1, r + 1), 3, cin, width)
            cin = width
    conv('fc6', 7 if variant == 'full' else 3, 512, c6)
    conv('fc7', 1, c6, c6)
    for i, layer in enumerate(FEAT_LAYERS):
        L = 'reverse_module/%s_reverse' % layer
        conv(L + '_conv_left', 2 if i == 0 else 3, c6 if i < 2 else 512, 512, bias=False)
        bn(L + '_conv_left', 512)
        if i > 0:
            out.append(('%s/%s_deconv_right/weights' % (SCOPE, L), (2, 2, 512, 512)))   # [kh,kw,Cout,Cin]
            out.append(('%s/%s_deconv_right/biases' % (SCOPE, L), (512,)))
        conv(L + '_objectness', 3, 512, 512, bias=False)
        bn(L + '_objectness', 512)
        conv(L + '_objectness_score', 3, 512, 2 * num_anchors)
        for blk, ic in ((1, 512), (2, 1024)):
            conv('%s_inception%d/Branch_0/Conv2d_3x3' % (L, blk), 3, ic, 512)
            conv('%s_inception%d/Branch_1/Conv2d_1x1' % (L, blk), 1, ic, 512)
            bn('%s_inception%d' % (L, blk), 1024)
        conv(L + '_inception2/Conv2d_pred_3x3', 3, 1024, num_anchors * num_classes)
        conv(L + '/Conv2d_0_3x3', 3, 512, 512, bias=False)
        bn(L + '/Conv2d_0_3x3', 512)
        conv(L + '/Conv2d_1_3x3', 3, 512, 4 * num_anchors)
    return out


SSD_SCOPE = 'ssd_512_vgg'
SSD_FEAT_LAYERS = ['block4', 'block7', 'block8', 'block9', 'block10', 'block11', 'block12']
SSD_ANCHORS = [4, 6, 6, 6, 6, 4, 4]
SSD_FEAT_CHANNELS = [512, 1024, 512, 256, 256, 256, 256]


def ssd_variable_shapes(num_classes=21):
    """(tf_name, shape) of SSD-512 (nets/ssd_vgg_512.py:364-460, multibox layer nets/ssd_vgg_300.py:403-431)."""
    out = []

    def conv(scope, k, cin, cout):
        out.append(('%s/%s/weights' % (SSD_SCOPE, scope), (k, k, cin, cout)))
        out.append(('%s/%s/biases' % (SSD_SCOPE, scope), (cout,)))

    cin = 3
    for b, (reps, width) in enumerate([(2, 64), (2, 128), (3, 256), (3, 512), (3, 512)]):
        for r in range(reps):
            conv('conv%d/conv%d_%d' % (b + 1, b + 1, r + 1), 3, cin, width)
            cin = width
    conv('conv6', 3, 512, 1024)
    conv('conv7', 1, 1024, 1024)
    for b, (inc, mid, outc) in enumerate([(1024, 256, 512), (512, 128, 256), (256, 128, 256), (256, 128, 256), (256, 128, 256)]):
        conv('block%d/conv1x1' % (8 + b), 1, inc, mid)
        conv('block%d/%s' % (8 + b, 'conv4x4' if b == 4 else 'conv3x3'), 4 if b == 4 else 3, mid, outc)
    for i, layer in enumerate(SSD_FEAT_LAYERS):
        if i == 0:
            out.append(('%s/%s_box/L2Normalization/gamma' % (SSD_SCOPE, layer), (512,)))
        conv(layer + '_box/conv_loc', 3, SSD_FEAT_CHANNELS[i], SSD_ANCHORS[i] * 4)
        conv(layer + '_box/conv_cls', 3, SSD_FEAT_CHANNELS[i], SSD_ANCHORS[i] * num_classes)
    return out


def ssd_synthetic_weights(num_classes=21, seed=5, bg=6.0, input_scale=1.0 / 64.0):
    """Seeded He-normal weights for SSD-512; +bg on every background logit; L2-norm scale 20 (the reference's init)."""
    rs = np.random.RandomState(seed)
    w = {}
    for name, shape in ssd_variable_shapes(num_classes):
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'weights':
            a = rs.standard_normal(int(np.prod(shape))).astype(np.float32).reshape(shape)
            a *= np.float32(np.sqrt(2.0 / (shape[0] * shape[1] * shape[2])))
            if name.endswith('conv1/conv1_1/weights'):
                a *= np.float32(input_scale)
            if '_box/' in name:
                a *= np.float32(0.3 / np.sqrt(2.0))           # logit std ~1 (block4's input is L2-normalised to norm 20)
        elif leaf == 'biases':
            a = (rs.standard_normal(shape) * 0.01).astype(np.float32)
            if name.endswith('conv_cls/biases'):
                a.reshape(-1, num_classes)[:, 0] += np.float32(bg)
        elif leaf == 'gamma':
            a = np.full(shape, 20.0, dtype=np.float32)
        else:
            raise AssertionError(name)
        w[name] = a
    return w


def synthetic_weights(variant='reducedfc', num_classes=21, num_anchors=10, seed=1, bg=8.0, ob=-4.0,
                      input_scale=1.0 / 64.0, head_gain=0.25):
    """Seeded random weights of the exact architecture.

    He-normal filters (std sqrt(2/fan_in)), BN gamma~U(.5,1.5), beta~N(0,.1), mean~N(0,.1), var~U(.5,1.5),
    biases N(0,.01).  conv1_1 is scaled by `input_scale` so activations of 0..255 pixel inputs stay O(1);
    the prediction layers get +bg on the background logit and +ob on the objectness-positive logit so that a
    realistic fraction of the 425 000 candidate scores passes the 0.01 threshold; `head_gain` brings the
    logit standard deviation to about 1 on every scale (measured: 0.8-1.4).
    """
    rs = np.random.RandomState(seed)
    w = {}
    for name, shape in variable_shapes(variant, num_classes, num_anchors):
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'weights':
            if '_deconv_right' in name:
                fan_in = shape[3]                    # one tap contributes to each output pixel
            else:
                fan_in = shape[0] * shape[1] * shape[2]
            a = rs.standard_normal(int(np.prod(shape))).astype(np.float32).reshape(shape)
            a *= np.float32(np.sqrt(2.0 / fan_in))
            if name.endswith('conv1/conv1_1/weights'):
                a *= np.float32(input_scale)
            if '_deconv_right' in name or ('_conv_left' in name and 'block7' not in name):
                a *= np.float32(np.sqrt(0.5))                 # the two summands of a reverse connection
            if name.endswith('Conv2d_pred_3x3/weights') or name.endswith('_objectness_score/weights') or \
                    name.endswith('Conv2d_1_3x3/weights'):
                a *= np.float32(head_gain / np.sqrt(2.0))     # linear layers: unit-variance logits
        elif leaf == 'biases':
            a = (rs.standard_normal(shape) * 0.01).astype(np.float32)
            if name.endswith('Conv2d_pred_3x3/biases'):
                a.reshape(num_anchors, num_classes)[:, 0] += np.float32(bg)
            if name.endswith('_objectness_score/biases'):
                a.reshape(num_anchors, 2)[:, 1] += np.float32(ob)
        elif leaf == 'gamma':
            a = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf in ('beta', 'moving_mean'):
            a = (rs.standard_normal(shape) * 0.1).astype(np.float32)
        elif leaf == 'moving_variance':
            a = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        else:
            raise AssertionError(name)
        w[name] = a
    return w


def synthetic_images(n, seed=0, img_shape=(320, 320)):
    """Pre-whitened synthetic batch: uniform 0..255 RGB minus the VGG means
    (preprocessing/ssd_vgg_preprocessing.py:30-32,376-377)."""
    rs = np.random.RandomState(seed)
    x = rs.uniform(0, 255, (n, img_shape[0], img_shape[1], 3)).astype(np.float32)
    return x - np.array([123., 117., 104.], dtype=np.float32)


def save_npz(path, weights):
    np.savez(path, **{k.replace('/', '|'): v for k, v in weights.items()})


def load_npz(path):
    with np.load(path) as z:
        return {k.replace('|', '/'): z[k] for k in z.files}
