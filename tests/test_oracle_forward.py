"""The conv-stack oracle against an independent implementation (CPU only).

``oracle/ron_forward.py`` / ``oracle/ssd_forward.py`` restate TensorFlow-1 / slim graph code
(``/root/reference/nets/ron_vgg_320.py:378-432,454-483,553-556``, ``nets/ssd_vgg_512.py:364-460``) that cannot run here
(parity unpinned).  What CAN be checked is that the restatement's own arithmetic -- the numpy im2col back-end -- computes
the layers it says it computes: the same networks are run once more with every conv / transposed conv / pool replaced
by the torch-CPU operator, and every intermediate tensor of every layer shape is compared.  The primitive ops are also
compared one by one on the padding / stride / dilation cases the two networks contain (SAME with odd and even kernels,
kernel == stride, pad2d + VALID stride 2, pad2d + 4x4 VALID, rates 3 and 6)."""
import numpy as np
import pytest

from oracle import ron_forward as orf
from oracle import ssd_forward as osf
from ron_tensorflow_amd import weights as W


def _rel(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize('k,rate,h,w', [(1, 1, 6, 5), (3, 1, 7, 9), (3, 3, 10, 10), (3, 6, 16, 16), (7, 1, 10, 10),
                                        (4, 1, 9, 9), (2, 1, 8, 6), (4, 1, 3, 3)])
def test_same_conv_backends_agree(k, rate, h, w):
    """SAME, stride 1: odd kernels (3x3, 7x7, rates 3 / 6) and even ones (TF pads the odd pixel after the data)."""
    rs = np.random.RandomState(k * 10 + rate)
    x = rs.randn(2, h, w, 5).astype(np.float32)
    wt = rs.randn(k, k, 5, 7).astype(np.float32)
    a, b = orf.conv2d_np(x, wt, 1, rate), orf.conv2d_torch(x, wt, 1, rate)
    assert a.shape == (2, h, w, 7)
    assert _rel(a, b) < 1e-5


def test_even_kernel_same_padding_is_after_the_data():
    """Hand case: a 2x2 all-ones kernel, SAME, stride 1 on a 2x2 map reads (y..y+1, x..x+1): bottom / right are padding."""
    x = np.arange(1, 5, dtype=np.float32).reshape(1, 2, 2, 1)
    w = np.ones((2, 2, 1, 1), np.float32)
    want = np.array([[10, 6], [7, 4]], np.float32).reshape(1, 2, 2, 1)
    assert np.array_equal(orf.conv2d_np(x, w), want) and np.array_equal(orf.conv2d_torch(x, w), want)


def test_strided_and_transposed_backends_agree():
    rs = np.random.RandomState(3)
    x = rs.randn(2, 10, 10, 6).astype(np.float32)
    w2 = rs.randn(2, 2, 6, 4).astype(np.float32)
    assert _rel(orf.conv2d_np(x, w2, 2), orf.conv2d_torch(x, w2, 2)) < 1e-5                 # block7 conv_left: 2x2 stride 2
    wt = rs.randn(2, 2, 4, 6).astype(np.float32)                                            # [kh, kw, Cout, Cin]
    a, b = orf.conv2d_transpose_np(x, wt), orf.conv2d_transpose_torch(x, wt)
    assert a.shape == (2, 20, 20, 4) and _rel(a, b) < 1e-5
    assert np.array_equal(orf.max_pool2x2_np(x), orf.max_pool2x2_torch(x))
    # SSD: pad2d(1) + 3x3 stride 2 VALID, pad2d(1) + 4x4 VALID, 3x3 stride-1 SAME max-pool
    w3 = rs.randn(3, 3, 6, 4).astype(np.float32)
    a, b = osf.conv2d_pad_np(x, w3, 2, 1, 1), osf.conv2d_pad_torch(x, w3, 2, 1, 1)
    assert a.shape == (2, 5, 5, 4) and _rel(a, b) < 1e-5
    w4 = rs.randn(4, 4, 6, 4).astype(np.float32)
    a, b = osf.conv2d_pad_np(x[:, :2, :2], w4, 1, 1, 1), osf.conv2d_pad_torch(x[:, :2, :2], w4, 1, 1, 1)
    assert a.shape == (2, 1, 1, 4) and _rel(a, b) < 1e-5
    assert np.array_equal(osf.max_pool3x3_s1_np(x), osf.max_pool3x3_s1_torch(x))


@pytest.mark.parametrize('variant', ['reducedfc', 'full'])
def test_ron_forward_backends_agree_on_every_layer(variant):
    """Every layer shape of ron_net / ron_net_reducedfc at 320 x 320: VGG body, fc6 (7x7 or rate 3), fc7, and per scale
    conv_left (2x2 s2 / 3x3), deconv, objectness, both inception blocks, box hidden, the three predictors."""
    weights = W.synthetic_weights(variant, seed=4)
    x = W.synthetic_images(1, seed=5)
    cn, ct = {}, {}
    on = orf.ron_forward(x, weights, variant, backend='numpy', collect=cn)
    ot = orf.ron_forward(x, weights, variant, backend='torch', collect=ct)
    assert set(cn) == set(ct) and len(cn) >= 13 + 5 + 4 * 5
    for name in cn:
        assert _rel(cn[name], ct[name]) < 2e-5, name
    for name in on[5]:
        assert _rel(on[5][name], ot[5][name]) < 2e-5, name
    for k in (1, 3, 4):                       # logits, objectness logits, localisations
        for a, b in zip(on[k], ot[k]):
            assert _rel(a, b) < 2e-5
    for k in (0, 2):                          # softmax outputs
        for a, b in zip(on[k], ot[k]):
            assert np.abs(a - b).max() < 1e-5


def test_ssd_forward_backends_agree_on_every_layer():
    """SSD-512: conv6 rate 6, pool5 3x3 s1, blocks 8-11 pad2d + stride-2 VALID, block12 4x4 VALID, multibox heads."""
    weights = W.ssd_synthetic_weights(seed=6)
    x = W.synthetic_images(1, seed=7, img_shape=(512, 512))
    cn, ct = {}, {}
    on = osf.ssd_forward(x, weights, collect=cn, backend='numpy')
    ot = osf.ssd_forward(x, weights, collect=ct, backend='torch')
    assert set(cn) == set(ct)
    for name in cn:
        assert _rel(cn[name], ct[name]) < 2e-5, name
    for name in on[3]:
        assert _rel(on[3][name], ot[3][name]) < 2e-5, name
    for k in (1, 2):
        for a, b in zip(on[k], ot[k]):
            assert _rel(a, b) < 2e-5
