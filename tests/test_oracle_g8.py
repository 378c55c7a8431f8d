"""The conv-stack oracle against the REFERENCE's own executable conv arithmetic (CPU only): golden G8.

``/root/reference/convert_pytorch_vgg.py:13-58`` defines the SSD-style vgg16_reducedfc backbone in plain torch (``VGG16``,
``vgg(cfg, i)``: 13 convs, pools M M C M, pool5 3x3 stride 1, conv6 3x3 dilation 6, conv7 1x1) - the same layers as
``nets/ssd_vgg_512.py:364-400`` up to block7 and as conv1_1 .. conv5_3 + pool1..4 of ``nets/ron_vgg_320.py:454-475``.
``tests/golden/make_golden.py::g8_vgg_backbone`` ran it in the build container on seeded weights (converted OIHW -> HWIO like
``nets/caffe_scope.py:57-60``) and seeded 320^2 / 512^2 images and stored every module's output (samples + whole-tensor sums).
Here the two restated oracles, on both of their back-ends, must reproduce those tensors.  What G8 cannot pin (no such layer in
the reference's torch model): fc6 7x7 / rate 3, pool5 2x2, the reverse-connection module, the heads, the transposed conv.  BatchNorm IS
pinned (second part of G8: the same reference function with batch_norm=True, conv -> BatchNorm2d -> ReLU in eval mode)."""
import numpy as np
import pytest

from oracle import ron_forward as orf
from oracle import ssd_forward as osf
from oracle import synth

from g8_util import G8, check_tensor


def test_g8_weight_generator_matches_the_stored_seed():
    assert int(G8['seed_weights']) == 80 and int(G8['seed_image_320']) == 81 and int(G8['seed_image_512']) == 82
    w = synth.vgg_backbone_weights_tf(80, 'ssd_512_vgg')
    assert w['ssd_512_vgg/conv1/conv1_1/weights'].shape == (3, 3, 3, 64)
    assert w['ssd_512_vgg/conv6/weights'].shape == (3, 3, 512, 1024) and w['ssd_512_vgg/conv7/weights'].shape == (1, 1, 1024, 1024)


@pytest.mark.parametrize('backend', ['numpy', 'torch'])
@pytest.mark.parametrize('size', [320, 512])
def test_ssd_oracle_backbone_reproduces_the_reference_vgg(size, backend):
    """oracle/ssd_forward.py conv1_1 .. conv7 (incl. pool5 3x3 s1 and the rate-6 conv6) == the reference's torch VGG16."""
    weights = synth.vgg_backbone_weights_tf(int(G8['seed_weights']), osf.SCOPE)
    img = synth.vgg_backbone_image(int(G8['seed_image_%d' % size]), size)
    collect = {}
    osf.ssd_forward(img, weights, collect=collect, backend=backend, stop_after='block7')
    assert sorted(collect) == sorted(synth.VGG_TAPS)
    worst = max(check_tensor(size, name, collect[name]) for name in synth.VGG_TAPS)
    print('ssd oracle (%s) vs G8 at %d: worst %.2e' % (backend, size, worst))


@pytest.mark.parametrize('backend', ['numpy', 'torch'])
def test_ron_oracle_vgg_body_reproduces_the_reference_vgg(backend):
    """oracle/ron_forward.py conv1_1 .. conv5_3 + pool1..4 (the code ron_forward itself runs first) == the reference's torch VGG16."""
    weights = synth.vgg_backbone_weights_tf(int(G8['seed_weights']), orf.SCOPE)
    img = synth.vgg_backbone_image(int(G8['seed_image_320']), 320)
    collect = orf.vgg_body(img, weights, backend=backend)
    names = [n for n in synth.VGG_TAPS if n not in ('pool5', 'conv6', 'conv7')]   # RON's pool5 is 2x2 s2, its fc6 / fc7 differ
    worst = max(check_tensor(320, name, collect[name]) for name in names)
    print('ron oracle (%s) vs G8: worst %.2e' % (backend, worst))


def test_g8_check_catches_a_wrong_border():
    """The check itself: one wrong border pixel of one channel (not among the sampled positions) must fail the sums."""
    weights = synth.vgg_backbone_weights_tf(int(G8['seed_weights']), orf.SCOPE)
    img = synth.vgg_backbone_image(int(G8['seed_image_320']), 320)
    net = orf._Net(weights, 'torch')
    a = net.conv_bias(img, 'conv1/conv1_1')
    check_tensor(320, 'conv1_1', a)
    b = a.copy()
    b[0, 5, 0, :] = 0          # row 5 is not sampled
    with pytest.raises(AssertionError):
        check_tensor(320, 'conv1_1', b)


@pytest.mark.parametrize('backend', ['numpy', 'torch'])
def test_oracle_conv_batchnorm_relu_reproduces_the_reference(backend):
    """slim.conv2d with normalizer_fn=slim.batch_norm at inference (no bias, BN eps 1e-5, ReLU: nets/ron_vgg_320.py:595-629) as the oracle
    computes it (oracle/ron_forward.py::_Net.conv_bn_relu) == the reference's own conv -> BatchNorm2d -> ReLU (vgg(cfg, i,
    batch_norm=True), convert_pytorch_vgg.py:47-48, eval mode, eps 1e-5), layer by layer, with the 2x2 pool between them."""
    params = synth.vgg_bn_params(int(G8['bn/seed_params']))
    weights = {}
    for i, (w, gamma, beta, mean, var) in enumerate(params):
        sc = '%s/bn_layer%d' % (orf.SCOPE, i)
        weights[sc + '/weights'] = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))          # OIHW -> HWIO, nets/caffe_scope.py:57-60
        weights[sc + '/BatchNorm/gamma'], weights[sc + '/BatchNorm/beta'] = gamma, beta
        weights[sc + '/BatchNorm/moving_mean'], weights[sc + '/BatchNorm/moving_variance'] = mean, var
    net = orf._Net(weights, backend)
    x = synth.vgg_backbone_image(int(G8['bn/seed_image']), 24)
    a1 = net.conv_bn_relu(x, 'bn_layer0')
    p1 = net.pool(a1)
    a2 = net.conv_bn_relu(p1, 'bn_layer1')
    for got, key in ((a1, 'bn/conv_bn_relu_1'), (p1, 'bn/pool'), (a2, 'bn/conv_bn_relu_2')):
        want = G8[key]
        assert got.shape == want.shape
        assert float(np.abs(got - want).max()) <= 1e-5 * float(np.abs(want).max()), key
