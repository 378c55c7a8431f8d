"""GPU parity: implicit-GEMM conv / transposed conv / max-pool kernels vs the oracle's numpy ops.

fp32 mode: exact-fp32 MFMA, tolerance 2e-5 relative to the output scale.  bf16/f16 modes: the oracle
gets the same operand rounding (inputs and weights rounded to the storage type, fp32 accumulate) and
the comparison allows one storage-type rounding of the output."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import ron_forward as orf  # noqa: E402

# f16x3 = split precision (two f16 planes per value, three f16 MFMAs per product): graded like fp32
ROUND = {'fp32': lambda a: np.asarray(a, np.float32), 'bf16': orf.round_bf16, 'fp16': orf.round_f16, 'f16x3': orf.round_f16x3}
OUT_EPS = {'fp32': 2e-5, 'bf16': 2 ** -7, 'fp16': 2 ** -10, 'f16x3': 2e-5}


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


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16', 'f16x3'])
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


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'f16x3'])
@pytest.mark.parametrize('cfg', [7, 9])
@pytest.mark.parametrize('shape', [(2, 10, 10, 64, 256, 3, 1, 3), (1, 10, 10, 64, 256, 7, 1, 1), (2, 10, 10, 128, 256, 2, 2, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_taps_innermost_orders_with_stride_rate_and_large_filters(ops, dev, shape, cfg, dtype):
    """The chunk-major / taps-innermost K order of tile configurations 7 (256 x 256) and 9 (128 x 128) only re-sequences the
    (tap, channel chunk) steps: rate 3, 7 x 7 and 2 x 2 / stride 2 filters against the oracle."""
    n, h, w, cin, cout, k, stride, rate = shape
    rs = np.random.RandomState(hash(shape) % 1000)
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt), stride, rate) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, stride=stride, dilation=rate, relu=True, dtype=dtype, tile_cfg=cfg,
                          splitk=1).cpu().numpy()
    assert got.shape == ref.shape
    _check(got, ref, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'f16x3'])
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


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16', 'f16x3'])
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


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'f16x3'])
def test_conv2d_transpose_2x2_with_residual(ops, dev, dtype):
    """The reverse connection as the graph runs it since round 4 (nets/ron_vgg_320.py:420-425): the 2x2 stride-2 transposed conv adds
    its half to the left conv's half in the pixel-shuffle epilogue, relu(relu(deconv + b) + left)."""
    rs = np.random.RandomState(41)
    x = rs.randn(3, 5, 5, 128).astype(np.float32)
    wt = (rs.randn(2, 2, 128, 128) * 0.08).astype(np.float32)        # [kh, kw, Cout, Cin]
    b = (rs.randn(128) * 0.1).astype(np.float32)
    left = np.maximum(rs.randn(3, 10, 10, 128), 0).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(np.maximum(orf.conv2d_transpose_np(rnd(x), rnd(wt), 2) + b, 0) + rnd(left), 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, residual=torch.from_numpy(left).to(dev), stride=2, relu=True,
                          transpose=True, dtype=dtype).cpu().numpy()
    assert got.shape == (3, 10, 10, 128)
    _check(got, ref, dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'f16x3'])
@pytest.mark.parametrize('cfg', [0, 7, 2, 9])
def test_centre_tap_only_columns_in_both_k_orders(ops, dev, cfg, dtype):
    """A 1x1 branch packed beside 3x3 ones (ConvLaunch::center_from; nets/ron_vgg_320.py:378-397): output channels >= center_from have
    weights in the centre tap only and their column tiles run that tap's K steps alone.  Round 4: also in the taps-innermost K order
    (tile configurations 7 and 9), where those tiles keep the tap-major walk of their one tap."""
    rs = np.random.RandomState(70 + cfg)
    n, h, w, cin, cout, cf = 5, 20, 20, 128, 512, 256
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(3, 3, cin, cout) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    centre = wt[1, 1, :, cf:].copy()
    wt[:, :, :, cf:] = 0
    wt[1, 1, :, cf:] = centre * 3                      # a 1x1 filter in the centre tap of its rows
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=cfg, splitk=1, center_from=cf).cpu().numpy()
    _check(got, ref, dtype)
    full = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=cfg, splitk=1).cpu().numpy()
    _check(got, full, dtype)          # the short tiles drop products with exact zeros only


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16', 'f16x3'])
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


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16', 'f16x3'])
def test_maxpool(ops, dev, dtype):
    rs = np.random.RandomState(8)
    x = ROUND[dtype](rs.randn(2, 8, 12, 64).astype(np.float32))
    got = ops.maxpool2x2_nhwc(torch.from_numpy(x).to(dev), dtype=dtype).cpu().numpy()
    assert np.array_equal(got, orf.max_pool2x2_np(x))


IGEMM_CFGS = [0, 1, 2, 3, 7, 9]     # conv_mfma.h: 256x256, 128x128, 128x128 early-issue, 128x64, 256x256 / 128x128 taps-innermost
PATCH_CFGS = {256: 4, 128: 5, 64: 6}  # halo-patch kernel by N tile


@pytest.mark.parametrize('dtype', ['bf16', 'fp32', 'f16x3'])
@pytest.mark.parametrize('cfg', IGEMM_CFGS)
def test_every_tile_configuration(ops, dev, cfg, dtype):
    """Each tile configuration of the row-gather kernel on a ragged multi-tile problem, K = 18 steps.  The shipped library
    holds exactly the configurations conv_pick_cfg() can select (the ablation builds live in libron_hip_diag.so)."""
    from ron_tensorflow_amd import _lib
    assert _lib.lib().ron_conv_num_tile_cfgs() == 11
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


def test_no_entry_accepts_a_configuration_it_cannot_run(ops, dev):
    """ron_conv2d_nhwc rejects tile configurations outside the selectable set, and the patch kernel on a conv it does not
    cover, instead of producing something."""
    from ron_tensorflow_amd._lib import RonError
    x = torch.zeros((1, 10, 10, 64), device=dev)
    w3 = np.zeros((3, 3, 64, 64), np.float32)
    for cfg in (10, 31, 100):
        with pytest.raises(RonError):
            ops.conv2d_nhwc(x, w3, None, dtype='bf16', tile_cfg=cfg)
    with pytest.raises(RonError):                                        # 1x1 conv through the 3x3 patch kernel
        ops.conv2d_nhwc(x, np.zeros((1, 1, 64, 64), np.float32), None, dtype='bf16', tile_cfg=6)
    with pytest.raises(RonError):                                        # N tile 256 on 64 output channels
        ops.conv2d_nhwc(x, w3, None, dtype='bf16', tile_cfg=4)
    with pytest.raises(RonError):                                        # resident-weight kernel: the 8 x 32 tile does not divide 10 x 10
        ops.conv2d_nhwc(x, w3, None, dtype='bf16', tile_cfg=8)
    with pytest.raises(RonError):                                        # ... and it has no fp32 form
        ops.conv2d_nhwc(torch.zeros((1, 8, 32, 64), device=dev), w3, None, dtype='fp32', tile_cfg=8)
    with pytest.raises(RonError):                                        # split precision: row-gather and halo-patch kernels, not this one
        ops.conv2d_nhwc(torch.zeros((1, 8, 32, 64), device=dev), w3, None, dtype='f16x3', tile_cfg=8)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'f16x3'])
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


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'f16x3'])
@pytest.mark.parametrize('cfg', [-1, 0, 1, 3, 7, 9])
def test_conv_with_fused_maxpool(ops, dev, dtype, cfg):
    """conv3x3 + bias + ReLU + 2x2/2 max-pool in one kernel == pool(conv) (nets/ron_vgg_320.py:454-466)."""
    rs = np.random.RandomState(70)
    cout = 64 if cfg == 3 else 256
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


@pytest.mark.parametrize('dtype,shape', [('fp32', (2, 320, 320, 64, 64)), ('bf16', (2, 40, 40, 512, 40)), ('fp32', (2, 40, 40, 512, 20)),
                                         ('f16x3', (2, 40, 40, 512, 40))])
def test_patch_kernel_first_step_waits_for_all_its_weights(ops, dev, dtype, shape):
    """The row-step form of the halo-patch kernel (N tile 64) once entered its first step with the weights of taps 1 and 2
    possibly still in flight: the compiler had merged the prologue's three placeholder LDS-DMA instructions into one and the
    counted wait came out two short.  It showed as run-to-run differences of up to 13 % on conv1_2 in fp32 at batch 2 (800
    workgroups), nowhere else in the suite.  Same launch eight times: identical bits, and equal to the row-gather kernel."""
    n, h, w, cin, cout = shape
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.randn(n, h, w, cin).astype(np.float32)).to(dev)
    wt = (rs.randn(3, 3, cin, cout) * 0.05).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    ref = ops.conv2d_nhwc(x, wt, b, relu=True, dtype=dtype, tile_cfg=3)
    outs = [ops.conv2d_nhwc(x, wt, b, relu=True, dtype=dtype, tile_cfg=6) for _ in range(8)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    tol = 1e-5 if dtype in ('fp32', 'f16x3') else 1e-2       # bf16: the two kernels add in different orders before the output rounding
    assert float((outs[0] - ref).abs().max()) <= tol * float(ref.abs().max())


PATCH_SHAPES = [  # n, h, w, cin, cout
    (2, 40, 40, 64, 128),     # flat runs of 256 positions (41 per row with the shared halo), crossing rows and the two images
    (1, 20, 20, 128, 256),    # flat, N tile 256, two chunks
    (3, 10, 10, 64, 64),      # flat, three images inside one and a half runs
    (2, 5, 5, 128, 128),      # flat, everything in one run
    (3, 44, 40, 192, 20),     # flat, Cout 20 -> N tile 64, masked tail, three chunks
    (2, 16, 96, 64, 256),     # 16 x 16 pixel tiles
    (1, 80, 80, 64, 210),     # 16 x 16 tiles, Cout 210 -> masked tail
    (1, 8, 96, 128, 64),      # 8 x 32 tiles (H not a multiple of 16)
]


def _patch_cfg(cout):
    npad = -(-cout // 64) * 64 if cout <= 64 else -(-cout // 128) * 128
    return PATCH_CFGS[256 if npad % 256 == 0 else (128 if npad % 128 == 0 else 64)]


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16', 'f16x3'])
@pytest.mark.parametrize('shape', PATCH_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_patch_kernel_conv3x3(ops, dev, shape, dtype):
    """The halo-patch 3x3 kernel (csrc/conv_patch.hip), flat and pixel-tile modes, vs the oracle conv.  Split precision (f16x3, round 4):
    both planes of a value sit in the same 128-byte patch row, the hi step of a tap reads slots 0..3, the lo step slots 4..7; the
    row-step form (N tile 64) keeps three fragment register sets so that the lo step still has the tap's hi fragments."""
    n, h, w, cin, cout = shape
    rs = np.random.RandomState(sum(shape))
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(3, 3, cin, cout) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=_patch_cfg(cout)).cpu().numpy()
    _check(got, ref, dtype)
    # the row-gather kernel on the same problem (its K loop runs tap-major, this one chunk-major: same sum, other order)
    same = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=3 if cout <= 64 else 1, splitk=1)
    if dtype != 'fp32':
        _check(got, same.cpu().numpy(), dtype)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_patch_kernel_channel_slice_residual_and_pool(ops, dev, dtype):
    rs = np.random.RandomState(91)
    rnd = ROUND[dtype]
    x = rs.randn(2, 40, 40, 128).astype(np.float32)
    wt = (rs.randn(3, 3, 128, 128) * 0.03).astype(np.float32)
    b = (rs.randn(128) * 0.1).astype(np.float32)
    res = np.maximum(rs.randn(2, 40, 40, 128), 0).astype(np.float32)
    ref = np.maximum(np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0) + rnd(res), 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, residual=torch.from_numpy(res).to(dev), relu=True, dtype=dtype,
                          tile_cfg=5).cpu().numpy()
    _check(got, ref, dtype)
    # input = channels [64, 192) of a 320-channel tensor (the heads read slices of the per-scale concatenated tensor)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, residual=torch.from_numpy(res).to(dev), relu=True, dtype=dtype,
                          tile_cfg=5, in_cstride=320, in_coff=64).cpu().numpy()
    _check(got, ref, dtype)
    for (h, w) in ((16, 64), (32, 32), (8, 96)):
        x = rs.randn(2, h, w, 64).astype(np.float32)
        wt = (rs.randn(3, 3, 64, 128) * 0.05).astype(np.float32)
        ref = orf.max_pool2x2_np(np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0))
        got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=5, pool=True).cpu().numpy()
        assert got.shape == ref.shape
        _check(got, ref, dtype)


C64_SHAPES = [  # n, h, w, cout (Cin = 64): the 3x3 kernel with resident weights (csrc/conv_c64.hip), 8 x 32 pixel tiles
    (1, 8, 32, 64),       # one tile, one 64-channel slice
    (2, 16, 64, 128),     # conv2_1-like: two slices, 8 tiles
    (3, 24, 96, 128),     # 27 tiles
    (1, 40, 32, 256),     # four slices
    (40, 32, 64, 128),    # 320 tiles on 128 tile sequences: every workgroup loops (double-buffered patches)
    (67, 8, 32, 64),      # 67 tiles on 72 sequences: some workgroups have nothing to do
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('shape', C64_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_resident_weight_kernel_conv3x3_c64(ops, dev, shape, dtype):
    """conv2_1's kernel vs the oracle conv, chosen explicitly (tile_cfg 8) and by conv_pick_cfg (-1): the same launch."""
    n, h, w, cout = shape
    rs = np.random.RandomState(sum(shape))
    x = rs.randn(n, h, w, 64).astype(np.float32)
    wt = (rs.randn(3, 3, 64, cout) * np.sqrt(2.0 / 576)).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    xd = torch.from_numpy(x).to(dev)
    got = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=8).cpu().numpy()
    _check(got, ref, dtype)
    auto = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype).cpu().numpy()
    assert np.array_equal(auto, got)
    # the row-gather kernel on the same problem: same sum in another order
    _check(got, ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=3 if cout == 64 else 1).cpu().numpy(), dtype)
    # no bias, no ReLU
    ref2 = orf.conv2d_np(rnd(x), rnd(wt))
    _check(ops.conv2d_nhwc(xd, wt, None, relu=False, dtype=dtype, tile_cfg=8).cpu().numpy(), ref2, dtype)


@pytest.mark.parametrize('dtype', ['bf16', 'fp32', 'f16x3'])
@pytest.mark.parametrize('cfg', [-1, 0, 1])
def test_weight_heavy_layer_runs_its_tiles_row_fastest(ops, dev, cfg, dtype):
    """fc6-like: few row tiles, several column tiles, K = 6272: without split-K conv_mfma.hip walks the tiles M fastest (the row tiles
    that share a weight slice share an XCD).  Same result as the oracle and as the split-K form (which keeps the N-fastest order)."""
    rs = np.random.RandomState(17)
    x = rs.randn(6, 10, 10, 128).astype(np.float32)             # M = 600: three 256-row / five 128-row tiles
    wt = (rs.randn(7, 7, 128, 512) * np.sqrt(2.0 / (49 * 128))).astype(np.float32)
    b = (rs.randn(512) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    xd = torch.from_numpy(x).to(dev)
    got = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=cfg, splitk=1).cpu().numpy()
    _check(got, ref, dtype)
    _check(ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=cfg, splitk=4).cpu().numpy(), ref, dtype)


# ---------------------------------------------------------------------------------------------------------------------
# split precision (RON_DTYPE_F16X3): what the three-MFMA product keeps
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('scale', [1.0, 2.0 ** -4, 300.0, 2.0 ** -10])
def test_split_precision_is_fp32_grade(ops, dev, scale):
    """A K = 4608 conv (conv4_x / the 3x3 heads) in f16x3 vs float64: error <= 4e-6 of the output scale (bf16: ~4e-3, f16: ~5e-4,
    the exact-fp32 MFMA mode: ~3e-7) at activation scales 1, 1/16 and 300 (the conv1 layers see +-130).
    The documented limit: a stored value carries an ABSOLUTE error floor of 2^-25 (the lo plane of |v| < 1/8 is an f16 subnormal,
    quantum 2^-24 -- which the matrix core does not flush), so a tensor whose whole scale is 2^-10 keeps ~15 bits: 2e-5 measured,
    still 20 x better than f16.  The networks of this repository run at activation scales >= 1."""
    rs = np.random.RandomState(11)
    x = (rs.randn(2, 12, 12, 512) * scale).astype(np.float32)
    wt = (rs.randn(3, 3, 512, 128) * np.sqrt(2.0 / 4608)).astype(np.float32)
    b = (rs.randn(128) * 0.1 * scale).astype(np.float32)
    xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)
    wt64 = torch.from_numpy(wt.astype(np.float64)).permute(3, 2, 0, 1)
    ref = (torch.nn.functional.conv2d(xt, wt64, padding=1).permute(0, 2, 3, 1).numpy() + b.astype(np.float64))
    errs = {}
    for dtype in ('f16x3', 'fp32', 'fp16'):
        got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=False, dtype=dtype).cpu().numpy()
        errs[dtype] = float(np.abs(got - ref).max() / np.abs(ref).max())
    print('scale %g: max err / output scale %s' % (scale, errs))
    assert errs['f16x3'] <= (4e-6 if scale >= 2.0 ** -4 else 5e-5), errs
    assert errs['f16x3'] <= 0.1 * errs['fp16'], errs


@pytest.mark.parametrize('exp', [-12, 14, 18])
def test_split_precision_outside_its_22_bit_range(ops, dev, exp):
    """The documented limits of dtype f16x3 (include/ron_hip.h, "VALID ACTIVATION RANGE") against float64, on inputs scaled by 2^exp:
    2^-12: every lo plane is an f16 subnormal -> absolute floor 2^-25 per stored value: the error is bounded by the floor summed
           over the K products (in quadrature: it is a rounding error), far better than f16 whose hi plane is itself subnormal there;
    2^+14: N(0,1) * 16384 -- a few inputs exceed 65504: they saturate (hi = 65504, lo = the rest) instead of becoming inf - inf = NaN,
           and the result stays fp32-grade;
    2^+18: everything beyond 131008 clips: wrong by the clipping, but finite (no NaN / inf reaches the next layer)."""
    rs = np.random.RandomState(13)
    scale = 2.0 ** exp
    x = (rs.randn(2, 12, 12, 64) * scale).astype(np.float32)
    wt = (rs.randn(3, 3, 64, 64) * np.sqrt(2.0 / 576) * (2.0 ** -4 if exp > 0 else 1.0)).astype(np.float32)    # outputs stay < 65504 at 2^14
    ref = orf.conv2d_np(x.astype(np.float64), wt.astype(np.float64))
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, None, relu=False, dtype='f16x3').cpu().numpy()
    assert np.isfinite(got).all(), 'a value outside the f16 range turned into inf / NaN'
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    print('f16x3 at input scale 2^%d: max err / output scale %.3g' % (exp, err))
    if exp == -12:
        floor = 2.0 ** -25 * np.sqrt(576) * float(np.abs(wt).max()) * 4     # per-value floor x sqrt(K) x |w|, with margin
        assert np.abs(got - ref).max() <= floor + 4e-6 * np.abs(ref).max(), err
        f16 = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, None, relu=False, dtype='fp16').cpu().numpy()
        assert err <= 0.25 * float(np.abs(f16 - ref).max() / np.abs(ref).max())
    elif exp == 14:
        assert (np.abs(x) > 65504).sum() > 0 and np.abs(x).max() < 131008
        assert err <= 2e-5, err           # the few saturated inputs keep 12 bits of their rest, the others 22
    else:
        assert np.abs(got).max() <= 9 * 64 * 131008 * float(np.abs(wt).max())   # clipped inputs bound the output


def test_split_precision_weight_scale_is_exact(ops, dev):
    """Weights far below / above 1 (the per-layer power-of-two scale brings them to [2^14, 2^15) and the epilogue undoes it): same
    relative error as at unit scale.  The activations are scaled the other way so that input and output stay inside what an
    f16 hi plane holds (|v| < 65504) and well above the 2^-25 absolute floor of the lo plane."""
    rs = np.random.RandomState(12)
    x0 = rs.randn(1, 10, 10, 64).astype(np.float32)
    for wscale in (2.0 ** -12, 2.0 ** -6, 1.0, 2.0 ** 4):
        x = (x0 / wscale).astype(np.float32)
        wt = (rs.randn(3, 3, 64, 64) * wscale * 0.05).astype(np.float32)
        ref = orf.conv2d_np(x.astype(np.float64), wt.astype(np.float64))
        got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, None, relu=False, dtype='f16x3').cpu().numpy()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err <= 4e-6, (wscale, err)


# ---------------------------------------------------------------------------------------------------------------------
# position-major rows + skipping of filter rows that only see the zero halo (ConvArgs::pos_major)
# ---------------------------------------------------------------------------------------------------------------------
HALO_SHAPES = [  # n, h, w, cin, cout, k, rate: small maps, many images -> a tile covers few output rows
    (32, 10, 10, 64, 256, 7, 1),     # fc6-like: 13 row tiles of 256, 4..7 of the 7 filter rows each
    (32, 5, 5, 128, 256, 3, 1),      # 3x3 on 5 x 5: the first and last output row skip a filter row
    (32, 10, 10, 64, 128, 3, 3),     # fc6 reduced: rate 3, rows 0-2 / 7-9 skip a filter row
    (24, 10, 10, 128, 210, 3, 1),    # masked N tail, ragged last tile
    (16, 16, 16, 64, 256, 3, 6),     # rate 6 on a 16 x 16 map (conv6 of SSD-512 in small)
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp32', 'f16x3'])
@pytest.mark.parametrize('shape', HALO_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_halo_filter_rows_are_skipped_not_missed(ops, dev, shape, dtype):
    """Launches whose rows the library orders position-major (a tile = the same few output positions of many images) and whose
    tiles skip the filter rows that fall into the zero halo for all of their rows -- alone, with every split-K factor (slices share
    what is left of the tile's K range; some are empty), with a residual, in every row-gather tile -- against the oracle conv."""
    n, h, w, cin, cout, k, rate = shape
    rs = np.random.RandomState(sum(shape))
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    res = np.maximum(rs.randn(n, h, w, cout), 0).astype(np.float32)
    rnd = ROUND[dtype]
    conv = orf.conv2d_np(rnd(x), rnd(wt), 1, rate) + b
    ref = np.maximum(conv, 0)
    ref_res = np.maximum(ref + rnd(res), 0)
    xd = torch.from_numpy(x).to(dev)
    cfgs = (-1, 0, 1) if cout % 256 == 0 else (-1, 1)        # tile 0 = 256 x 256
    for cfg in cfgs:
        for splitk in (1, 2, 5, -1):
            got = ops.conv2d_nhwc(xd, wt, b, dilation=rate, relu=True, dtype=dtype, tile_cfg=cfg, splitk=splitk).cpu().numpy()
            _check(got, ref, dtype)
    got = ops.conv2d_nhwc(xd, wt, b, residual=torch.from_numpy(res).to(dev), dilation=rate, relu=True, dtype=dtype, splitk=1).cpu().numpy()
    _check(got, ref_res, dtype)
    got = ops.conv2d_nhwc(xd, wt, b, residual=torch.from_numpy(res).to(dev), dilation=rate, relu=True, dtype=dtype, splitk=3).cpu().numpy()
    _check(got, ref_res, dtype)


@pytest.mark.parametrize('dtype', ['bf16', 'fp16', 'f16x3'])
def test_tile_256x128_of_the_assembly_loop(ops, dev, dtype):
    """Tile configuration 10 (conv_mfma.h kCfgIgemm256x128: 256 x 128 on four waves, the assembly K loop with 4 column tiles per
    wave; what conv_pick_cfg takes for conv2_x-sized layers with Cout = 128): ragged M over several tiles, Cout below the tile
    width, K of 18 / 1 / 2 / 5 steps (shorter than the two tiles the loop keeps in flight, odd counts), a residual, fp32 output."""
    rs = np.random.RandomState(90)
    rnd = ROUND[dtype]
    x = rs.randn(3, 13, 11, 128).astype(np.float32)                # M = 429: two 256-row tiles, the second ragged
    wt = (rs.randn(3, 3, 128, 100) * 0.03).astype(np.float32)      # Cout 100 -> padded to 128: the last lanes' vectors are partial
    b = (rs.randn(100) * 0.1).astype(np.float32)
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    xd = torch.from_numpy(x).to(dev)
    got = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=10).cpu().numpy()
    _check(got, ref, dtype)
    # the same launch in the 128 x 128 tile: same products, the K order may differ (taps innermost) -> equal within rounding
    _check(got, ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=1).cpu().numpy(), dtype)
    chunk = 32 if dtype == 'f16x3' else 64
    for steps in (1, 2, 5):
        x2 = rs.randn(2, 9, 15, chunk * steps).astype(np.float32)
        w2 = (rs.randn(1, 1, chunk * steps, 128) * 0.05).astype(np.float32)
        res = np.maximum(rs.randn(2, 9, 15, 128), 0).astype(np.float32)
        ref2 = np.maximum(orf.conv2d_np(rnd(x2), rnd(w2)) + rnd(res), 0)
        got2 = ops.conv2d_nhwc(torch.from_numpy(x2).to(dev), w2, None, residual=torch.from_numpy(res).to(dev), relu=False, dtype=dtype,
                               tile_cfg=10).cpu().numpy()
        _check(got2, ref2, dtype)
    # 256 x 128 is not a tile of the fp32 form
    with pytest.raises(Exception, match='256 x 128'):
        ops.conv2d_nhwc(xd, wt, b, relu=True, dtype='fp32', tile_cfg=10)


@pytest.mark.parametrize('dtype', ['bf16', 'f16x3'])
def test_tile_256x128_with_fused_maxpool_and_by_choice(ops, dev, dtype):
    """conv2_2's launch: 3x3, 128 -> 128, 2x2 max-pool fused, on a map large enough that conv_pick_cfg (-1) takes the 256 x 128 tile
    (>= 256 tiles): the chosen launch and the explicit one give the same bits, both equal pool(conv) of the oracle."""
    rs = np.random.RandomState(91)
    rnd = ROUND[dtype]
    x = rs.randn(2, 160, 208, 128).astype(np.float32)              # M = 66 560 rows = 260 tiles of 256
    wt = (rs.randn(3, 3, 128, 128) * 0.03).astype(np.float32)
    b = (rs.randn(128) * 0.1).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    got = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=10, pool=True).cpu().numpy()
    auto = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=-1, pool=True).cpu().numpy()
    assert np.array_equal(got, auto)
    ref = orf.max_pool2x2_np(np.maximum(orf.conv2d_np(rnd(x[:1, :32]), rnd(wt)) + b, 0))      # the oracle on the top strip of image 0
    _check(got[:1, :15], ref[:, :15], dtype)                        # (its last conv row lacks the strip's lower neighbour)
    full = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=10)
    assert np.array_equal(ops.maxpool2x2_nhwc(full, dtype=dtype).cpu().numpy(), got)


@pytest.mark.parametrize('dtype', ['bf16', 'f16x3'])
def test_tile_256x128_split_k_and_one_round_launches(ops, dev, dtype):
    """Tile configuration 10 beyond conv2_x: several column tiles (Cout 200 -> 256), K split over 2 / 3 / 5 slices (the slices' fp32 slabs
    and the finalize pass behind the four-wave tile), and the launches conv_pick_cfg now gives it by itself - about one round of the
    chip (48 .. 320 tiles of 256 x 128): the chosen launch equals the explicit one bit for bit."""
    rs = np.random.RandomState(92)
    rnd = ROUND[dtype]
    x = rs.randn(3, 13, 11, 128).astype(np.float32)                # M = 429: two 256-row tiles, the second ragged
    wt = (rs.randn(3, 3, 128, 200) * 0.03).astype(np.float32)      # Cout 200 -> 256: two column tiles, the second with partial vectors
    b = (rs.randn(200) * 0.1).astype(np.float32)
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    xd = torch.from_numpy(x).to(dev)
    one = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=10, splitk=1).cpu().numpy()
    _check(one, ref, dtype)
    for sk in (2, 3, 5, -1):
        got = ops.conv2d_nhwc(xd, wt, b, relu=True, dtype=dtype, tile_cfg=10, splitk=sk).cpu().numpy()
        _check(got, ref, dtype)
        _check(got, one, dtype)
    # a launch of 50 x 4 = 200 such tiles (conv5_x at batch 32: 20 x 20 x 32 rows, 512 -> 512): the picker's own choice
    x2 = rs.randn(32, 20, 20, 512).astype(np.float32) * 0.5
    w2 = (rs.randn(3, 3, 512, 512) * 0.02).astype(np.float32)
    b2 = (rs.randn(512) * 0.1).astype(np.float32)
    x2d = torch.from_numpy(x2).to(dev)
    auto = ops.conv2d_nhwc(x2d, w2, b2, relu=True, dtype=dtype, tile_cfg=-1).cpu().numpy()
    explicit = ops.conv2d_nhwc(x2d, w2, b2, relu=True, dtype=dtype, tile_cfg=10).cpu().numpy()
    assert np.array_equal(auto, explicit)
    ref2 = np.maximum(orf.conv2d_np(rnd(x2[:1]), rnd(w2)) + b2, 0)
    _check(auto[:1], ref2, dtype)
    # ... and with the fused 2x2 max-pool
    pooled = ops.conv2d_nhwc(x2d, w2, b2, relu=True, dtype=dtype, tile_cfg=-1, pool=True)
    assert np.array_equal(pooled.cpu().numpy(), ops.maxpool2x2_nhwc(torch.from_numpy(auto).to(dev), dtype=dtype).cpu().numpy())


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'f16x3'])
@pytest.mark.parametrize('splitk', [1, 3, -1])
@pytest.mark.parametrize('cfg', [-1, 1, 0, 10])
@pytest.mark.parametrize('heads', [(84, 16), (486, 24), (21, 4), (126, 24)], ids=lambda h: '%d+%d' % h)
def test_two_head_outputs_from_one_convolution(ops, dev, heads, cfg, splitk, dtype):
    """ConvArgs::split_n (the class + box convolutions of an SSD feature layer as one launch, nets/ssd_vgg_300.py:403-431; only the SSD-512
    graph reached it before): against two separate convolutions of the oracle, over tile configurations, forced split-K (the finalize
    pass routes the columns too) and class counts whose first head is NOT a multiple of 8 columns (81 classes x 6 anchors = 486,
    21 x 4 = 84, 21 x 1 = 21; 126 = 6 x 21)."""
    from ron_tensorflow_amd._lib import RonError
    n_cls, n_loc = heads
    if cfg == 10 and dtype == 'fp32':
        pytest.skip('the 256 x 128 tile is the assembly loop of the 16-bit types')
    rs = np.random.RandomState(n_cls + 7 * max(cfg, 0))
    cin = 128
    x = rs.randn(2, 8, 8, cin).astype(np.float32)
    wt = (rs.randn(3, 3, cin, n_cls + n_loc) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rs.randn(n_cls + n_loc) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = orf.conv2d_np(rnd(x), rnd(wt)) + b
    packed = -(-n_cls // 8) * 8 + n_loc                      # the second head starts at the next multiple of 8
    tile = 64 if packed <= 64 else 128                       # conv_n_tile
    npad = -(-packed // tile) * tile
    try:
        y1, y2 = ops.conv2d_heads_nhwc(torch.from_numpy(x).to(dev), wt, n_cls, bias=b, dtype=dtype, tile_cfg=cfg, splitk=splitk)
    except RonError:
        # a forced tile must divide the packed width (256-wide tiles: Npad % 256; the 128-wide ones: Npad % 128)
        assert (cfg == 0 and npad % 256 != 0) or (cfg in (1, 10) and npad % 128 != 0)
        return
    assert tuple(y1.shape) == (2, 8, 8, n_cls) and tuple(y2.shape) == (2, 8, 8, n_loc)
    _check(y1.cpu().numpy(), ref[..., :n_cls], dtype)
    _check(y2.cpu().numpy(), ref[..., n_cls:], dtype)


@pytest.mark.parametrize('dtype', ['bf16', 'f16x3'])
@pytest.mark.parametrize('case', [(1024, 0, 2), (768, 0, 2), (1280, 0, 2), (1024, 768, 2), (2048, 1536, 2)], ids=lambda c: 'n%d_c%d' % c[:2])
def test_panel_tile_orders_cover_every_tile(ops, dev, case, dtype):
    """Round 6: tile orders in panels of P column tiles (ConvArgs::m_fastest >= 2) - even and ragged last panels (1024 = 4, 768 = 3,
    1280 = 5 column tiles of 256) and the long columns of a launch with centre-tap-only columns.  The plan must say so (ron_conv_plan),
    and a tile order that skipped or repeated a tile could not reproduce the oracle."""
    cout, center_from, want_p = case
    n, h, w, cin = 16, 32, 32, 64                                 # M = 16 384: 64 row tiles of 256, an XCD's 32 resident workgroups are all of this launch
    plan = ops.conv_plan(n, h, w, cin, cout, k=3, dtype=dtype, tile_cfg=0, splitk=1, center_from=center_from)
    assert plan['tile_cfg'] == 0 and plan['splitk'] == 1 and plan['tile_order'] == want_p, plan
    rs = np.random.RandomState(cout + center_from)
    x = rs.randn(n, h, w, cin).astype(np.float32)
    wt = (rs.randn(3, 3, cin, cout) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    if center_from:
        wt[:, :, :, center_from:] = 0
        wt[1, 1, :, center_from:] = (rs.randn(cin, cout - center_from) * np.sqrt(2.0 / cin)).astype(np.float32)
    b = (rs.randn(cout) * 0.1).astype(np.float32)
    rnd = ROUND[dtype]
    ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt)) + b, 0)
    got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, relu=True, dtype=dtype, tile_cfg=0, splitk=1, center_from=center_from).cpu().numpy()
    _check(got, ref, dtype)
