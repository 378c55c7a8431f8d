"""GPU parity against the REFERENCE's own executable conv stack (golden G8, see tests/test_oracle_g8.py).

The device's fp32 and split-precision (f16x3) paths run the seeded weights and images of G8 through the C ABI
(ron_create / ron_load_weight / ron_forward / ron_end_point_copy) and every tensor the reference's torch VGG16
(/root/reference/convert_pytorch_vgg.py:13-58) produced is compared: RON-320 conv1_1 .. conv5_3 + pool1..4 at 320^2
(nets/ron_vgg_320.py:454-475), SSD-512 conv1_1 .. conv7 at 512^2 (nets/ssd_vgg_512.py:364-400: pool5 3x3 s1, conv6 rate 6).
Tolerance: 1e-4 of each tensor's largest value (north_star's float bound); achieved values are printed."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from g8_util import G8, check_tensor  # noqa: E402
from oracle import synth  # noqa: E402

TOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _ron(dtype, fuse_pools, dev):
    import ron_tensorflow_amd.weights as W
    from ron_tensorflow_amd.nets import nets_factory
    weights = W.synthetic_weights('reducedfc', seed=1)
    g8w = synth.vgg_backbone_weights_tf(int(G8['seed_weights']), W.SCOPE)
    for k, v in g8w.items():
        if '/conv6/' in k or '/conv7/' in k:
            continue                               # RON's fc6 / fc7 are other layers
        assert weights[k].shape == v.shape, k
        weights[k] = v
    cls = nets_factory.get_network('ron_320_vgg')
    net = cls(cls.default_params._replace(num_classes=21), variant='reducedfc', dtype=dtype, max_batch=1, device=dev,
              fuse_pools=fuse_pools)
    return net.load_weights(weights)


@pytest.mark.parametrize('dtype', ['fp32', 'f16x3'])
def test_ron_vgg_body_reproduces_the_reference_vgg(dtype, dev):
    net = _ron(dtype, False, dev)
    x = torch.from_numpy(synth.vgg_backbone_image(int(G8['seed_image_320']), 320)).to(dev)
    net.forward_heads(x)
    names = [n for n in synth.VGG_TAPS if n not in ('pool5', 'conv6', 'conv7')]
    errs = {n: check_tensor(320, n, net.end_point(n, 1).cpu().numpy(), tol=TOL, sum_tol=1e-5) for n in names}
    # the reference's end_points block1 .. block5 are the same tensors by their public names
    for b, n in (('block1', 'conv1_2'), ('block2', 'conv2_2'), ('block3', 'conv3_3'), ('block4', 'conv4_3'), ('block5', 'conv5_3')):
        assert torch.equal(net.end_point(b, 1), net.end_point(n, 1))
    net.close()
    print('RON-320 %s device path vs G8: worst %.2e (%s)' % (dtype, max(errs.values()), max(errs, key=errs.get)))


@pytest.mark.parametrize('dtype', ['fp32', 'f16x3'])
def test_ron_fused_pool_path_reproduces_the_reference_vgg(dtype, dev):
    """The benchmarked launch plan (pools in the conv epilogues; fused stem where the dtype has one): pool1..3, block4, block5."""
    net = _ron(dtype, True, dev)
    x = torch.from_numpy(synth.vgg_backbone_image(int(G8['seed_image_320']), 320)).to(dev)
    net.forward_heads(x)
    for n in ('pool1', 'pool2', 'pool3', 'conv4_1', 'conv4_3', 'pool4', 'conv5_3'):
        check_tensor(320, n, net.end_point(n, 1).cpu().numpy(), tol=TOL, sum_tol=1e-5)
    net.close()


@pytest.mark.parametrize('dtype', ['fp32', 'f16x3'])
def test_ssd_backbone_reproduces_the_reference_vgg(dtype, dev):
    import ron_tensorflow_amd.weights as W
    from ron_tensorflow_amd.nets import nets_factory
    weights = W.ssd_synthetic_weights(seed=5)
    for k, v in synth.vgg_backbone_weights_tf(int(G8['seed_weights']), W.SSD_SCOPE).items():
        assert weights[k].shape == v.shape, k
        weights[k] = v
    cls = nets_factory.get_network('ssd_512_vgg')
    net = cls(cls.default_params._replace(num_classes=21), dtype=dtype, max_batch=1).load_weights(weights)
    x = torch.from_numpy(synth.vgg_backbone_image(int(G8['seed_image_512']), 512)).to(dev)
    net.forward_heads(x)
    errs = {n: check_tensor(512, n, net.end_point(n, 1).cpu().numpy(), tol=TOL, sum_tol=1e-5) for n in synth.VGG_TAPS}
    assert torch.equal(net.end_point('block7', 1), net.end_point('conv7', 1))
    net.close()
    print('SSD-512 %s device path vs G8: worst %.2e (%s)' % (dtype, max(errs.values()), max(errs, key=errs.get)))
