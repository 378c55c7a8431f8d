"""GPU parity: implicit-GEMM conv / transposed conv / max-pool kernels vs the oracle's numpy ops.

fp32 mode: exact-fp32 MFMA, tolerance 2e-5 relative to the output scale.  bf16/f16 modes: the oracle
gets the same operand rounding (inputs and weights rounded to the storage type, fp32 accumulate) and
the comparison allows one storage-type rounding of the output."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import ron_forward as orf  # noqa: E402

ROUND = {'fp32': lambda a: np.asarray(a, np.float32), 'bf16': orf.round_bf16, 'fp16': orf.round_f16}
OUT_EPS = {'fp32': 2e-5, 'bf16': 2 ** -7, 'fp16': 2 ** -10}


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from ron_tensorflow_amd import ops as _ops
    return _ops


def _check(got, ref, dtype):
    scale = float(np.abs(ref).max()) + 1e-6
    err = np.abs(got - ref).max() / scale
    assert err <= OUT_EPS[dtype] * 1.5, 'max err / scale = %g' % err


CONV_SHAPES = [
    # n, h, w, cin, cout, k, stride, rate
    (2, 12, 10, 64, 64, 3, 1, 1),       # N tile 64
    (1, 9, 7, 128, 128, 3, 1, 1),       # N tile 128, ragged M
    (3, 10, 10, 64, 20, 3, 1, 1),       # objectness_score-like: Cout 20 -> padded rows masked
    (1, 10, 10, 128, 210, 3, 1, 1),     # cls pred
    (2, 10, 10, 192, 256, 1, 1, 1),     # 1x1
    (2, 10, 10, 64, 128, 3, 1, 3),      # fc6 reduced: rate 3
    (1, 10, 10, 64, 128, 7, 1, 1),      # fc6 full: 7x7
    (2, 10, 10, 128, 128, 2, 2, 1),     # block7 conv_left: 2x2 stride 2
    (1, 40, 40, 64, 64, 3, 1, 1),       # many tiles
]


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('shape', CONV_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_conv2d(ops, dev, shape, dtype):
    n, h, w, cin, cout, k, stride, rate = shape
    rs = np.random.RandomState(hash(shape) % 1000)
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt), stride, rate) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, stride=stride, dilation=rate, relu=True,
                          dtype=dtype).cpu().numpy()
    assert got.shape == ref.shape
    _check(got, ref, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_conv2d_no_relu_no_bias_and_residual(ops, dev, dtype):
    rs = np.random.RandomState(5)
    x = rs.randn(2, 10, 10, 128).astype(np.float32)
    wt = (rs.randn(3, 3, 128, 128) * 0.03).astype(np.float32)
    rnd = ROUND[dtype]
    ref = orf.conv2d_np(rnd(x), rnd(wt))
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, None, relu=False, dtype=dtype).cpu().numpy()
    assert (got < 0).any()
    _check(got, ref, dtype)
    # reverse-connection sum: relu(relu(conv + b) + residual)   (nets/ron_vgg_320.py:422-425)
    res = np.maximum(rs.randn(2, 10, 10, 128), 0).astype(np.float32)
    b = (rs.randn(128) * 0.1).astype(np.float32)
    ref = np.maximum(np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0) + rnd(res), 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, residual=torch.from_numpy(res).to(dev), relu=True,
                          dtype=dtype).cpu().numpy()
    _check(got, ref, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16'])
def test_conv2d_transpose_2x2(ops, dev, dtype):
    rs = np.random.RandomState(6)
    x = rs.randn(2, 5, 5, 128).astype(np.float32)
    wt = (rs.randn(2, 2, 128, 128) * 0.08).astype(np.float32)       # [kh, kw, Cout, Cin]
    b = (rs.randn(128) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_transpose_np(rnd(x), rnd(wt), 2) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, stride=2, relu=True, transpose=True,
                          dtype=dtype).cpu().numpy()
    assert got.shape == (2, 10, 10, 128)
    _check(got, ref, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('hw', [(16, 20), (12, 64), (7, 96)])
def test_conv_stem_3_channels(ops, dev, dtype, hw):
    """conv1_1: width % 32 == 0 goes through the dedicated stem kernel (bf16 / f16), otherwise im2col + 1x1 GEMM."""
    rs = np.random.RandomState(7)
    x = (rs.uniform(0, 255, (2, hw[0], hw[1], 3)) - np.array([123., 117., 104.])).astype(np.float32)
    wt = (rs.randn(3, 3, 3, 64) * np.sqrt(2.0 / 27)).astype(np.float32)
    b = (rs.randn(64) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype).cpu().numpy()
    _check(got, ref, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16'])
def test_maxpool(ops, dev, dtype):
    rs = np.random.RandomState(8)
    x = ROUND[dtype](rs.randn(2, 8, 12, 64).astype(np.float32))
    got = ops.maxpool2x2_nhwc(torch.from_numpy(x).to(dev), dtype=dtype).cpu().numpy()
    assert np.array_equal(got, orf.max_pool2x2_np(x))


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
@pytest.mark.parametrize('cfg', [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 16, 20, 21, 22, 23, 24, 26, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 51, 52, 53, 56, 57, 58, 59, 60, 61, 62])
def test_every_tile_configuration(ops, dev, cfg, dtype):
    """Each (tile, wave grid, stage count) variant of the kernel on a ragged multi-tile problem, K = 18 steps."""
    from ron_tensorflow_amd import _lib
    assert _lib.lib().ron_conv_num_tile_cfgs() == 64
    rs = np.random.RandomState(40 + cfg)
    x = rs.randn(3, 13, 11, 128).astype(np.float32)            # M = 429: two 256-row or four 128-row tiles, ragged
    wt = (rs.randn(3, 3, 128, 192) * 0.03).astype(np.float32)  # Cout 192 -> padded to 256
    b = (rs.randn(192) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=cfg).cpu().numpy()
    _check(got, ref, dtype)
    # K = 1 step and K = 2 steps (shorter than the pipeline depth)
    for cin, k in ((64, 1), (128, 1)):
        x2 = rs.randn(2, 9, 9, cin).astype(np.float32)
        w2 = (rs.randn(k, k, cin, 256) * 0.1).astype(np.float32)
        ref2 = orf.conv2d_np(rnd(x2), rnd(w2))
        got2 = ops.conv2d_nhwc(torch.from_numpy(x2).to(dev), w2, None, relu=False, dtype=dtype, tile_cfg=cfg).cpu().numpy()
        _check(got2, ref2, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('splitk', [1, 2, 3, 7, -1])
def test_split_k(ops, dev, dtype, splitk):
    """Small grid, long K (the 5x5 / 10x10 head layers): K loop spread over workgroups, slabs summed by a second kernel."""
    rs = np.random.RandomState(60)
    x = rs.randn(2, 5, 5, 256).astype(np.float32)
    rnd = ROUND[dtype]
    for cout, relu, with_res in ((20, False, False), (128, True, True), (210, False, False)):
        wt = (rs.randn(3, 3, 256, cout) * 0.03).astype(np.float32)
        b = (rs.randn(cout) * 0.1).astype(np.float32)
        res = np.maximum(rs.randn(2, 5, 5, cout), 0).astype(np.float32) if with_res else None
        ref = orf.conv2d_np(rnd(x), rnd(wt)) + b
        if relu:
            ref = np.maximum(ref, 0)
        if with_res:
            ref = np.maximum(ref + rnd(res), 0)
        got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, residual=None if res is None else torch.from_numpy(res).to(dev),
                              relu=relu, dtype=dtype, splitk=splitk).cpu().numpy()
        _check(got, ref, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('cfg', [-1, 0, 5, 6, 11])
def test_conv_with_fused_maxpool(ops, dev, dtype, cfg):
    """conv3x3 + bias + ReLU + 2x2/2 max-pool in one kernel == pool(conv) (nets/ron_vgg_320.py:454-466)."""
    rs = np.random.RandomState(70)
    cout = 64 if cfg == 5 else 256
    x = rs.randn(3, 12, 20, 64).astype(np.float32)
    wt = (rs.randn(3, 3, 64, cout) * 0.05).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = orf.max_pool2x2_np(np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0))
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=cfg, pool=True).cpu().numpy()
    assert got.shape == (3, 6, 10, cout)
    _check(got, ref, dtype)
    # same kernel configuration without split-K: fused pooling == pooling the stored conv output, bit for bit
    full = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=cfg, splitk=1)
    assert np.array_equal(ops.maxpool2x2_nhwc(full, dtype=dtype).cpu().numpy(), got)


PATCH_SHAPES = [  # n, h, w, cin, cout
    (2, 40, 40, 64, 128),     # six full 40-wide rows per tile, ragged last tile row
    (1, 12, 64, 128, 64),     # 4 x 64 strips, N tile 64
    (2, 16, 96, 64, 256),     # 8 x 32 tiles
    (1, 80, 80, 64, 210),     # 40-wide half rows, Cout 210 -> masked tail
    (3, 44, 40, 192, 20),     # Cout 20
]


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('shape', PATCH_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_patch_kernel_conv3x3(ops, dev, shape, dtype):
    """The halo-patch 3x3 kernel (csrc/conv_patch.hip, tile_cfg = 100) vs the oracle conv."""
    n, h, w, cin, cout = shape
    rs = np.random.RandomState(sum(shape))
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(3, 3, cin, cout) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=100).cpu().numpy()
    _check(got, ref, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_patch_kernel_residual_and_pool(ops, dev, dtype):
    rs = np.random.RandomState(91)
    rnd = ROUND[dtype]
    x = rs.randn(2, 40, 40, 128).astype(np.float32)
    wt = (rs.randn(3, 3, 128, 128) * 0.03).astype(np.float32)
    b = (rs.randn(128) * 0.1).astype(np.float32)
    res = np.maximum(rs.randn(2, 40, 40, 128), 0).astype(np.float32)
    ref = np.maximum(np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0) + rnd(res), 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, residual=torch.from_numpy(res).to(dev), relu=True, dtype=dtype,
                          tile_cfg=100).cpu().numpy()
    _check(got, ref, dtype)
    for (h, w) in ((40, 40), (16, 64), (24, 96)):
        x = rs.randn(2, h, w, 64).astype(np.float32)
        wt = (rs.randn(3, 3, 64, 128) * 0.05).astype(np.float32)
        ref = orf.max_pool2x2_np(np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0))
        got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=100, pool=True).cpu().numpy()
        assert got.shape == ref.shape
        _check(got, ref, dtype)
