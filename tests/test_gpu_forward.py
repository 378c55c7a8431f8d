"""GPU parity: the conv stack (ron_forward through RONNet) vs the oracle forward, and the fused detect path.

fp32 mode is the parity mode (exact-fp32 MFMA): head tensors within 1e-4 of the logit scale.  bf16/fp16
modes are compared with an oracle that applies the same operand rounding, with a tolerance of a few
storage-type ulps; their detections are graded on identical head tensors ("parity vs precision")."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import anchors as oanchors  # noqa: E402
from oracle import np_post  # noqa: E402
from oracle import ron_forward as orf  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def pkg():
    import ron_tensorflow_amd.weights as W
    from ron_tensorflow_amd.nets import nets_factory, ron_vgg_320
    return dict(W=W, factory=nets_factory, ron=ron_vgg_320)


@pytest.fixture(scope='module')
def weights_reduced(pkg):
    return pkg['W'].synthetic_weights('reducedfc', seed=1)


@pytest.fixture(scope='module')
def weights_full(pkg):
    return pkg['W'].synthetic_weights('full', seed=2)


@pytest.fixture(scope='module')
def images(pkg):
    return pkg['W'].synthetic_images(2, seed=0)


@pytest.fixture(scope='module')
def oracle_reduced(weights_reduced, images):
    col = {}
    out = orf.ron_forward(images, weights_reduced, 'reducedfc', backend='numpy', collect=col)
    return out, col


def _rel_err(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


def test_factory_interface(pkg):
    f = pkg['factory']
    cls = f.get_network('ron_320_vgg')
    assert cls is pkg['ron'].RONNet
    params = cls.default_params._replace(num_classes=21)
    assert params.img_shape == (320, 320) and params.feat_shapes[0] == (5, 5)
    with pytest.raises(ValueError):
        f.get_network_fn('no_such_net', 21)
    with pytest.raises(KeyError):
        f.get_network('no_such_net')


def test_variable_contract_and_flops(pkg, dev):
    for variant, gflop in (('reducedfc', 138.26), ('full', 164.16)):
        net = pkg['ron'].RONNet(variant=variant, dtype='fp32', max_batch=1)
        assert net.variables() == [(n, tuple(s)) for n, s in pkg['W'].variable_shapes(variant)]
        net.close()
    # FLOPs are known after finalize (needs weights): checked in the forward tests below


def test_forward_fp32_reducedfc(pkg, dev, weights_reduced, images, oracle_reduced):
    (pred, logits, objp, objl, loc, eps), col = oracle_reduced
    net = pkg['ron'].RONNet(variant='reducedfc', dtype='fp32', max_batch=2).load_weights(weights_reduced)
    assert abs(net.flops_per_image() / 1e9 - 138.26) < 0.5
    x = torch.from_numpy(images).to(dev)
    g_pred, g_logits, g_objp, g_objl, g_loc, g_eps = net.net(x, is_training=False)
    for i in range(4):
        assert tuple(g_logits[i].shape) == logits[i].shape and tuple(g_pred[i].shape) == pred[i].shape
        assert tuple(g_objp[i].shape) == objp[i].shape and tuple(g_loc[i].shape) == loc[i].shape
        assert _rel_err(g_logits[i].cpu().numpy(), logits[i]) < 1e-4, i
        assert _rel_err(g_objl[i].cpu().numpy(), objl[i]) < 1e-4, i
        assert _rel_err(g_loc[i].cpu().numpy(), loc[i]) < 1e-4, i
        np.testing.assert_allclose(g_pred[i].cpu().numpy(), pred[i], rtol=0, atol=1e-4)
        np.testing.assert_allclose(g_objp[i].cpu().numpy(), objp[i], rtol=0, atol=1e-4)
    for name in ('block1', 'block4', 'block5', 'block6', 'block7'):
        assert _rel_err(g_eps[name].cpu().numpy(), eps[name]) < 1e-4, name
    for name in ('block7_ref', 'block4_ref', 'pool5'):
        assert _rel_err(net.end_point(name, 2).cpu().numpy(), col[name]) < 1e-4, name

    # fused detect == post-processing of the heads the same context produced (identical head tensors)
    anchors = oanchors.anchors_all_layers()
    det = net.detect(x).to_lists()
    ref = np_post.detect_from_predictions([p.cpu().numpy() for p in g_pred], [l.cpu().numpy() for l in g_loc], anchors,
                                          objness_pred=[o.cpu().numpy() for o in g_objp])
    for b in range(2):
        assert np.array_equal(det[b]['classes'], ref[b]['classes'])
        assert np.array_equal(det[b]['anchor_index'], ref[b]['anchor_index'])
        assert np.array_equal(det[b]['scores'], ref[b]['scores'])
        np.testing.assert_allclose(det[b]['bboxes'], ref[b]['bboxes'], rtol=0, atol=1e-5)
    # fully independent oracle (own conv stack): detections agree except where a score / IoU sits within
    # float rounding of a threshold; require >= 98 % of the reference detections reproduced, floats to 1e-4
    ind = np_post.detect_from_predictions(pred, loc, anchors, objness_pred=objp)
    for b in range(2):
        ref_keys = {(int(c), int(a)): k for k, (c, a) in enumerate(zip(ind[b]['classes'], ind[b]['anchor_index']))}
        hit = [(k, ref_keys[(int(c), int(a))]) for k, (c, a) in enumerate(zip(det[b]['classes'], det[b]['anchor_index']))
               if (int(c), int(a)) in ref_keys]
        assert len(hit) >= 0.98 * len(ref_keys) and len(hit) >= 0.98 * len(det[b]['classes'])
        gi, ri = np.array([h[0] for h in hit]), np.array([h[1] for h in hit])
        np.testing.assert_allclose(det[b]['scores'][gi], ind[b]['scores'][ri], rtol=0, atol=1e-4)
        np.testing.assert_allclose(det[b]['bboxes'][gi], ind[b]['bboxes'][ri], rtol=0, atol=1e-4)
    net.close()


def test_forward_fp32_full(pkg, dev, weights_full, images):
    ref = orf.ron_forward(images[:1], weights_full, 'full', backend='numpy')
    net = pkg['ron'].RONNet(variant='full', dtype='fp32', max_batch=1).load_weights(weights_full)
    assert abs(net.flops_per_image() / 1e9 - 164.16) < 0.5
    cls, obj, loc = net.forward_heads(torch.from_numpy(images[:1]).to(dev))
    for i in range(4):
        assert _rel_err(cls[i].cpu().numpy(), ref[1][i]) < 1e-4, i
        assert _rel_err(obj[i].cpu().numpy(), ref[3][i]) < 1e-4, i
        assert _rel_err(loc[i].cpu().numpy(), ref[4][i]) < 1e-4, i
    net.close()


@pytest.mark.parametrize('fuse_pools', [False, True])
def test_forward_split_precision_reducedfc(pkg, dev, weights_reduced, images, oracle_reduced, fuse_pools):
    """dtype 'f16x3' (two f16 planes per value, three f16 MFMAs per product, fp32 accumulate) against the all-fp32 oracle, with
    the bounds of the exact-fp32 mode: every head tensor and end point within 1e-4 of its scale (measured ~2e-6), detections of
    the fused path >= 98 % those of the independent oracle with scores / boxes within 1e-4."""
    (pred, logits, objp, objl, loc, eps), col = oracle_reduced
    net = pkg['ron'].RONNet(variant='reducedfc', dtype='f16x3', max_batch=2, fuse_pools=fuse_pools).load_weights(weights_reduced)
    x = torch.from_numpy(images).to(dev)
    g_pred, g_logits, g_objp, g_objl, g_loc, g_eps = net.net(x, is_training=False)
    worst = 0.0
    for i in range(4):
        for g, r in ((g_logits[i], logits[i]), (g_objl[i], objl[i]), (g_loc[i], loc[i])):
            worst = max(worst, _rel_err(g.cpu().numpy(), r))
        np.testing.assert_allclose(g_pred[i].cpu().numpy(), pred[i], rtol=0, atol=1e-4)
        np.testing.assert_allclose(g_objp[i].cpu().numpy(), objp[i], rtol=0, atol=1e-4)
    for name in ('block4', 'block5', 'block6', 'block7'):
        worst = max(worst, _rel_err(g_eps[name].cpu().numpy(), eps[name]))
    for name in ('block7_ref', 'block4_ref', 'pool5'):
        worst = max(worst, _rel_err(net.end_point(name, 2).cpu().numpy(), col[name]))
    print('f16x3 reducedfc (fuse_pools=%s): worst head / end-point error relative to the tensor scale: %.3g' % (fuse_pools, worst))
    assert worst < 2e-5
    anchors = oanchors.anchors_all_layers()
    det = net.detect(x).to_lists()
    ind = np_post.detect_from_predictions(pred, loc, anchors, objness_pred=objp)
    for b in range(2):
        ref_keys = {(int(c), int(a)): k for k, (c, a) in enumerate(zip(ind[b]['classes'], ind[b]['anchor_index']))}
        hit = [(k, ref_keys[(int(c), int(a))]) for k, (c, a) in enumerate(zip(det[b]['classes'], det[b]['anchor_index']))
               if (int(c), int(a)) in ref_keys]
        assert len(hit) >= 0.98 * len(ref_keys) and len(hit) >= 0.98 * len(det[b]['classes'])
        gi, ri = np.array([h[0] for h in hit]), np.array([h[1] for h in hit])
        np.testing.assert_allclose(det[b]['scores'][gi], ind[b]['scores'][ri], rtol=0, atol=1e-4)
        np.testing.assert_allclose(det[b]['bboxes'][gi], ind[b]['bboxes'][ri], rtol=0, atol=1e-4)
    # deterministic
    again = net.forward_heads(x)
    for i in range(4):
        assert torch.equal(again[0][i], g_logits[i].reshape(again[0][i].shape))
    net.close()


def test_forward_split_precision_full(pkg, dev, weights_full, images):
    """ron_net full (fc6: K = 25 088) in f16x3 vs the fp32 oracle."""
    ref = orf.ron_forward(images[:1], weights_full, 'full', backend='numpy')
    net = pkg['ron'].RONNet(variant='full', dtype='f16x3', max_batch=1).load_weights(weights_full)
    cls, obj, loc = net.forward_heads(torch.from_numpy(images[:1]).to(dev))
    for i in range(4):
        assert _rel_err(cls[i].cpu().numpy(), ref[1][i]) < 2e-5, i
        assert _rel_err(obj[i].cpu().numpy(), ref[3][i]) < 2e-5, i
        assert _rel_err(loc[i].cpu().numpy(), ref[4][i]) < 2e-5, i
    net.close()


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_forward_reduced_precision(pkg, dev, weights_reduced, images, oracle_reduced, dtype):
    rnd = {'bf16': orf.round_bf16, 'fp16': orf.round_f16}[dtype]
    tol = {'bf16': 0.04, 'fp16': 0.006}[dtype]
    ref = orf.ron_forward(images[:1], weights_reduced, 'reducedfc', backend='numpy', round_fn=rnd)
    net = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=2).load_weights(weights_reduced)
    x = torch.from_numpy(images).to(dev)
    cls, obj, loc = net.forward_heads(x)
    for i in range(4):
        g = cls[i][:1].cpu().numpy()
        assert _rel_err(g, ref[1][i]) < tol, (i, _rel_err(g, ref[1][i]))
        assert float(np.abs(g - ref[1][i]).mean() / np.abs(ref[1][i]).max()) < tol / 8
        assert _rel_err(loc[i][:1].cpu().numpy(), ref[4][i]) < tol * 2
    # detections of the reduced-precision stack are graded on its own head tensors (bit-exact post-processing)
    from ron_tensorflow_amd import ops
    anchors = oanchors.anchors_all_layers()
    pred = [ops.softmax_last(c) for c in cls]
    objp = [ops.softmax_last(o, pick=1) for o in obj]
    want = np_post.detect_from_predictions([p.cpu().numpy() for p in pred], [l.cpu().numpy() for l in loc], anchors,
                                           objness_pred=[o.cpu().numpy() for o in objp])
    det = net.detect(x).to_lists()
    for b in range(2):
        assert np.array_equal(det[b]['classes'], want[b]['classes'])
        assert np.array_equal(det[b]['anchor_index'], want[b]['anchor_index'])
        assert np.array_equal(det[b]['scores'], want[b]['scores'])
        np.testing.assert_allclose(det[b]['bboxes'], want[b]['bboxes'], rtol=0, atol=1e-5)
    # same batch twice -> bitwise identical (no atomics anywhere: split-K sums slabs in a fixed order)
    again = net.forward_heads(x)
    for i in range(4):
        assert torch.equal(again[0][i], cls[i]) and torch.equal(again[2][i], loc[i])
    # image 1 alone vs inside the batch: the split-K factor follows the grid size, so the fp32 summation order (and
    # with it a few storage-type roundings) may differ between batch sizes -- equal within the dtype tolerance
    one = net.forward_heads(x[1:2])
    for i in range(4):
        a, b = one[0][i][0].cpu().numpy(), cls[i][1].cpu().numpy()
        assert float(np.abs(a - b).max() / np.abs(b).max()) < tol
    net.close()


def test_fused_pools_give_identical_heads(pkg, dev, weights_reduced, images):
    """RON_CFG_FUSE_POOLS only changes where the max-pool runs.  max commutes with the monotonic bias / ReLU / rounding
    steps, so fused and separate pooling agree bit for bit per element; at this tiny batch the un-fused conv3_3 runs
    split-K (a different fp32 summation order), hence a few bf16 ulps instead of bitwise equality."""
    x = torch.from_numpy(images).to(dev)
    a = pkg['ron'].RONNet(variant='reducedfc', dtype='bf16', max_batch=2).load_weights(weights_reduced)
    b = pkg['ron'].RONNet(variant='reducedfc', dtype='bf16', max_batch=2, fuse_pools=True).load_weights(weights_reduced)
    ha, hb = a.forward_heads(x), b.forward_heads(x)
    for la, lb in zip(ha, hb):
        for ta, tb in zip(la, lb):
            ta, tb = ta.cpu().numpy(), tb.cpu().numpy()
            assert float(np.abs(ta - tb).max() / np.abs(ta).max()) < 0.04
            assert float(np.abs(ta - tb).mean() / np.abs(ta).max()) < 0.002
    ea, eb = a.end_point('block4', 2).cpu().numpy(), b.end_point('block4', 2).cpu().numpy()
    assert float(np.abs(ea - eb).max() / np.abs(ea).max()) < 0.04
    with pytest.raises(Exception):
        b.end_point('block1', 2)
    out = b.net(x, is_training=False)
    assert sorted(out[5]) == ['block4', 'block5', 'block6', 'block7']
    a.close()
    b.close()


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_conv4_3_writes_its_map_and_its_pool_in_one_launch(pkg, dev, weights_reduced, dtype):
    """With fused pools conv4_3 / conv5_3 store BOTH the full-resolution map (read by the reverse-connection conv,
    nets/ron_vgg_320.py:495-506) and the pooled one from the same accumulators whenever the launch does not split K (batch 8:
    conv4_3 does not, conv5_3 still does and keeps its pool launch).  Same tiles, same K order as the separate launches: the maps
    are bit-identical, and so is everything downstream."""
    x = torch.from_numpy(pkg['W'].synthetic_images(8, seed=31)).to(dev)
    a = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=8, fuse_pools=True).load_weights(weights_reduced)
    b = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=8, fuse_pools=False).load_weights(weights_reduced)
    ha, hb = a.forward_heads(x), b.forward_heads(x)
    for name in ('block4', 'pool4', 'block5', 'pool5'):
        ea, eb = a.end_point(name, 8), b.end_point(name, 8)
        assert float(eb.abs().max()) > 0
        if dtype == 'fp32':
            assert torch.equal(ea, eb), name
        else:       # bf16: the fused stem / block1-3 pools reorder fp32 partial sums upstream
            assert float((ea - eb).abs().max()) <= 0.04 * float(eb.abs().max()), name
    if dtype == 'fp32':
        for la, lb in zip(ha, hb):
            for ta, tb in zip(la, lb):
                assert float((ta - tb).abs().max()) <= 2e-5 * float(tb.abs().max())
    a.close()
    b.close()


@pytest.mark.parametrize('dtype,tol', [('fp32', 2e-5), ('bf16', 2e-2)])
def test_halo_filter_row_skipping_changes_nothing(pkg, dev, weights_reduced, dtype, tol):
    """RON_CFG_NO_HALO_SKIP: image-major rows and the full K loop everywhere.  The default (position-major rows on the small maps,
    tiles skip the filter rows that fall outside the image for all of their rows) only drops products with halo zeros; where K is
    split the slices cut what is left of the range, so fp32 partial sums are grouped differently -- equal to rounding."""
    x = torch.from_numpy(pkg['W'].synthetic_images(8, seed=41)).to(dev)
    a = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=8).load_weights(weights_reduced)
    b = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=8)
    b.no_halo_skip = True
    b.load_weights(weights_reduced)
    ha, hb = a.forward_heads(x), b.forward_heads(x)
    for la, lb in zip(ha, hb):
        for ta, tb in zip(la, lb):
            assert float((ta - tb).abs().max()) <= tol * float(tb.abs().max())
    for name in ('block6', 'block7', 'block7_ref', 'block6_ref'):
        ea, eb = a.end_point(name, 8), b.end_point(name, 8)
        assert float((ea - eb).abs().max()) <= tol * float(eb.abs().max()), name
    a.close()
    b.close()


def test_multi_stream_heads_are_identical(pkg, dev, weights_reduced, images):
    """RON_CFG_MULTI_STREAM only changes which stream a head branch is enqueued on: bitwise the same tensors,
    also when calls follow each other without a host sync (fork/join ordering)."""
    x = torch.from_numpy(images).to(dev)
    # one launch per convolution on both sides (side streams do not combine with grouped launches): identical launches, identical bits
    a = pkg['ron'].RONNet(variant='reducedfc', dtype='bf16', max_batch=2, group_heads=False).load_weights(weights_reduced)
    b = pkg['ron'].RONNet(variant='reducedfc', dtype='bf16', max_batch=2, multi_stream=True).load_weights(weights_reduced)
    ha = a.forward_heads(x)
    for _ in range(3):
        hb = b.forward_heads(x)
    x2 = torch.flip(x, dims=[0]).contiguous()
    hb2 = b.forward_heads(x2)                 # back-to-back call with different data, no sync in between
    for la, lb, lb2 in zip(ha, hb, hb2):
        for ta, tb, tb2 in zip(la, lb, lb2):
            assert torch.equal(ta, tb)
            assert torch.equal(ta, torch.flip(tb2, dims=[0]))
    da, db = a.detect(x).to_lists(), b.detect(x).to_lists()
    for u, v in zip(da, db):
        for k in ('classes', 'scores', 'bboxes', 'anchor_index'):
            assert np.array_equal(u[k], v[k])
    a.close()
    b.close()


@pytest.mark.parametrize('variant', ['reducedfc', 'full'])
def test_grouped_launch_plan_is_active(pkg, dev, weights_reduced, weights_full, variant):
    """The grouped head launches are keyed on op names inside libron_hip (plan_groups): a rename in the graph builder would
    silently fall back to one launch per convolution (-7 % of the step) while every equivalence test still passed.  Read the plan."""
    w = weights_reduced if variant == 'reducedfc' else weights_full
    net = pkg['ron'].RONNet(variant=variant, dtype='bf16', max_batch=24, fuse_pools=True).load_weights(w)
    plan = net.launch_plan()
    groups = [n for n in plan if n.startswith('group[')]
    assert net.grouped_launches() == 10 and len(groups) == 10, plan
    # round 4: the left conv of a reverse connection reads a backbone map only and is off the coarse -> fine chain: the left convs are
    # the carriers of conv5_1 and fc7 (256 x 256 groups), and the small convolutions of a dependency level ride in the partial
    # rounds of a large one or share a mixed-width launch
    assert groups == ['group[conv5_1+1]', 'group[fc7+1]', 'group[block7_conv_left+1]', 'group[block7_trio3+1]', 'group[block6_trio3+2]',
                      'group[block5_trio3+3]', 'group[block7_objectness_score+5]', 'group[block4_trio3+2]',
                      'group[block4_objectness_score+1]', 'group[block4_inception2+1]'], groups
    assert plan[0] == 'conv1_1+conv1_2+pool1' and plan[-1] == 'post_np'
    launches = [n for n in plan[:-1] if not n.startswith('(')]
    assert len(launches) == 25, (len(launches), launches)       # fused stem + 9 launches up to pool4 + 15 from conv5_1 on
    mid = pkg['ron'].RONNet(variant=variant, dtype='bf16', max_batch=16, fuse_pools=True).load_weights(w)      # 12 < max_batch < 24
    assert [n for n in mid.launch_plan() if n.startswith('group[')] == [
        'group[fc7+1]', 'group[block7_conv_left+1]', 'group[block7_trio3+1]', 'group[block7_objectness_score+4]',
        'group[block7_cls_pred+5]', 'group[block6_cls_pred+3]', 'group[block4_objectness_score+1]']
    mid.close()
    clone = net.clone()
    assert clone.launch_plan() == plan and clone.grouped_launches() == 10
    clone.close()
    net.close()
    net = pkg['ron'].RONNet(variant=variant, dtype='bf16', max_batch=1, fuse_pools=True, group_heads=False).load_weights(w)
    assert net.grouped_launches() == 0 and not any(n.startswith('group[') for n in net.launch_plan())
    net.close()
    # small contexts (max_batch <= 12): the heads go out one launch per dependency level, the large convolutions grouped too
    net = pkg['ron'].RONNet(variant=variant, dtype='bf16', max_batch=4, fuse_pools=True).load_weights(w)
    plan = net.launch_plan()
    groups = [n for n in plan if n.startswith('group[')]
    assert groups == ['group[block7_conv_left+3]', 'group[block7_trio3+1]', 'group[block7_objectness_score+4]', 'group[block7_cls_pred+5]',
                      'group[block6_cls_pred+4]', 'group[block5_cls_pred+3]'], groups
    assert len([n for n in plan[:-1] if not n.startswith('(')]) == 23 and net.grouped_launches() == 6      # 7 head launches
    net.close()
    forced = pkg['ron'].RONNet(variant=variant, dtype='bf16', max_batch=4, fuse_pools=True, head_plan='batch').load_weights(w)
    assert len([n for n in forced.launch_plan()[:-1] if not n.startswith('(')]) == 27
    forced.close()
    forced = pkg['ron'].RONNet(variant=variant, dtype='bf16', max_batch=8, fuse_pools=True, head_plan='level').load_weights(w)
    assert len([n for n in forced.launch_plan()[:-1] if not n.startswith('(')]) == 23
    forced.close()


@pytest.mark.parametrize('dtype,tol', [('fp32', 2e-5), ('bf16', 2e-2), ('f16x3', 2e-5)])
def test_level_plan_matches_batch_plan(pkg, dev, weights_reduced, images, dtype, tol):
    """RON_CFG_LEVEL_GROUPS vs RON_CFG_BATCH_GROUPS: the same convolutions in other launches (tiles and the split of K differ, i.e.
    the order of the fp32 partial sums); both deterministic."""
    x = torch.from_numpy(images).to(dev)
    a = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=2, fuse_pools=True, head_plan='level').load_weights(weights_reduced)
    b = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=2, fuse_pools=True, head_plan='batch').load_weights(weights_reduced)
    ha, hb = a.forward_heads(x), b.forward_heads(x)
    for ta, tb in zip(ha, hb):
        for u, v in zip(ta, tb):
            assert _rel_err(u.cpu().numpy(), v.cpu().numpy()) <= tol
    for name in ('block7_ref', 'block6_ref', 'block5_ref', 'block4_ref'):
        assert _rel_err(a.end_point(name, 2).cpu().numpy(), b.end_point(name, 2).cpu().numpy()) <= tol, name
    ha2 = a.forward_heads(x)
    for ta, tb in zip(ha, ha2):
        for u, v in zip(ta, tb):
            assert torch.equal(u, v)
    a.close()
    b.close()


def test_network_fn_uses_full_variant(pkg, dev, weights_full, images):
    fn = pkg['factory'].get_network_fn('ron_320_vgg', 21, is_training=False, weights=weights_full, dtype='bf16', max_batch=1)
    assert fn.default_image_size == 320
    out = fn(torch.from_numpy(images[:1]).to(dev), end_points=('block6',))
    assert len(out) == 6
    assert tuple(out[5]['block6'].shape) == (1, 10, 10, 4096)       # ron_net: fc6 7x7 -> 4096 (nets/ron_vgg_320.py:478)
    assert tuple(out[0][3].shape) == (1, 40, 40, 10, 21)


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_fused_stem_matches_separate_launches(pkg, dev, weights_reduced, dtype):
    """conv1_1 + conv1_2 + pool1 as one kernel (stem2_kernel) vs stem kernel -> conv kernel with fused pool: pool1 agrees to
    one storage-type ulp (different fp32 accumulation order inside conv1_2), heads to the reduced-precision tolerance."""
    cls = pkg['factory'].get_network('ron_320_vgg')
    x = torch.from_numpy(pkg['W'].synthetic_images(3, seed=12)).to(dev)
    outs = []
    for no_stem2 in (False, True):
        net = cls(variant='reducedfc', dtype=dtype, max_batch=3, device=dev, fuse_pools=True)
        net.no_stem2 = no_stem2
        net.load_weights(weights_reduced)
        lg, _, _ = net.forward_heads(x)
        outs.append((net.end_point('pool1', 3).cpu().numpy(), [t.cpu().numpy() for t in lg]))
        net.close()
    p_a, p_b = outs[0][0], outs[1][0]
    assert p_a.shape == (3, 160, 160, 64) and np.abs(p_b).max() > 0
    ulp = 2.0 ** (-7 if dtype == 'bf16' else -10)
    assert np.abs(p_a - p_b).max() <= 1.01 * ulp * np.abs(p_b).max()
    assert np.mean(p_a != p_b) < 0.05                               # and nearly everywhere bit-identical
    for a, b in zip(outs[0][1], outs[1][1]):
        assert _rel_err(a, b) < (3e-2 if dtype == 'bf16' else 5e-3)


@pytest.mark.parametrize('dtype,tol', [('fp32', 2e-5), ('bf16', 3e-2), ('f16x3', 2e-5)])
def test_ragged_batch_sizes(pkg, dev, weights_reduced, dtype, tol):
    """Every batch size up to max_batch goes through the same launch plan (tile counts, split-K factors and the fused stem's tile
    loop change with n): the first n images of a batch of 7 give the heads of a batch of n."""
    cls = pkg['factory'].get_network('ron_320_vgg')
    net = cls(variant='reducedfc', dtype=dtype, max_batch=7, device=dev, fuse_pools=dtype == 'bf16')
    net.load_weights(weights_reduced)
    x = torch.from_numpy(pkg['W'].synthetic_images(7, seed=21)).to(dev)
    full = [t.clone() for t in net.forward_heads(x)[0]]
    for n in (1, 2, 3, 5, 6):
        part = net.forward_heads(x[:n])[0]
        for a, b in zip(part, full):
            assert torch.isfinite(a).all()
            scale = float(b[:n].abs().max())
            assert float((a - b[:n]).abs().max()) <= tol * scale, (n, dtype)
    with pytest.raises(Exception):
        net.forward_heads(torch.cat([x, x[:1]]))                  # 8 > max_batch
    net.close()


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'f16x3'])
def test_forward_is_bitwise_deterministic(pkg, dev, weights_reduced, dtype):
    """Same input, same context: the same bits, run after run, for every end point and head tensor, at several batch sizes of a
    small and a medium context (both grouped plans, split-K at every level, the halo-patch and row-gather kernels).  A counted
    LDS-DMA wait that came out short once showed as nothing but run-to-run differences of block1 in fp32 at batch 2; the parity
    tests of this file passed or failed by chance over it."""
    names = ('block1', 'block2', 'block3', 'block4', 'block5', 'block6', 'block7', 'block7_ref', 'block6_ref', 'block5_ref', 'block4_ref')
    for mb in (2, 7):
        net = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=mb).load_weights(weights_reduced)
        x = torch.from_numpy(pkg['W'].synthetic_images(mb, seed=21)).to(dev)
        for n in sorted({1, 2, mb}):
            ref = None
            for _ in range(5):
                heads = net.forward_heads(x[:n])
                cur = [t.clone() for grp in heads if grp is not None for t in grp] + [net.end_point(nm, n).clone() for nm in names]
                if ref is None:
                    ref = cur
                for i, (a, b) in enumerate(zip(cur, ref)):
                    assert torch.equal(a, b), (dtype, mb, n, i)
        net.close()


@pytest.mark.parametrize('dtype,tol', [('fp32', 2e-5), ('bf16', 2e-2), ('f16x3', 2e-5)])
def test_grouped_head_launches_match_one_launch_per_conv(pkg, dev, weights_reduced, images, dtype, tol):
    """RON_CFG_NO_GROUPS: the same graph with every head convolution as its own launch.  Grouping changes tiles and the split
    of K, i.e. only the order of fp32 partial sums (and, in bf16, where a value sits relative to a rounding boundary of the
    next layer's input)."""
    x = torch.from_numpy(images).to(dev)
    a = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=2).load_weights(weights_reduced)
    b = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=2, group_heads=False).load_weights(weights_reduced)
    ha, hb = a.forward_heads(x), b.forward_heads(x)
    for ta, tb in zip(ha, hb):
        for u, v in zip(ta, tb):
            assert _rel_err(u.cpu().numpy(), v.cpu().numpy()) <= tol
    for name in ('block7_ref', 'block6_ref', 'block5_ref', 'block4_ref'):
        assert _rel_err(a.end_point(name, 2).cpu().numpy(), b.end_point(name, 2).cpu().numpy()) <= tol, name
    # grouped launches are deterministic: same bits on a second run
    ha2 = a.forward_heads(x)
    for ta, tb in zip(ha, ha2):
        for u, v in zip(ta, tb):
            assert torch.equal(u, v)
    a.close()
    b.close()


@pytest.mark.parametrize('max_batch,n', [(24, 24), (16, 13)])
@pytest.mark.parametrize('dtype,tol', [('fp32', 2e-5), ('f16x3', 2e-5)])
def test_carrier_and_mid_plans_match_one_launch_per_conv(pkg, dev, weights_reduced, max_batch, n, dtype, tol):
    """The launch plans of larger contexts (csrc/graph.cpp plan_groups: max_batch >= 24 packs the small head convolutions into the
    partial rounds of 256 x 256 launches - kCfgIgemm256 groups, entries ordered long tiles first -; 13..23 runs a dependency level as
    one mixed-width launch) against RON_CFG_NO_GROUPS at a batch that fills them, in the arithmetic where equality is tight; twice, for
    determinism (split-K factors of a group come from a schedule model and are cached per batch)."""
    x = torch.from_numpy(pkg['W'].synthetic_images(n, seed=77)).to(dev)
    a = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=max_batch, fuse_pools=True).load_weights(weights_reduced)
    b = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=max_batch, fuse_pools=True, group_heads=False).load_weights(weights_reduced)
    assert a.grouped_launches() == (10 if max_batch >= 24 else 7) and b.grouped_launches() == 0
    ha, hb = a.forward_heads(x), b.forward_heads(x)
    for ta, tb in zip(ha, hb):
        for u, v in zip(ta, tb):
            assert _rel_err(u.cpu().numpy(), v.cpu().numpy()) <= tol
    for name in ('block7', 'block7_ref', 'block6_ref', 'block5_ref', 'block4_ref'):
        assert _rel_err(a.end_point(name, n).cpu().numpy(), b.end_point(name, n).cpu().numpy()) <= tol, name
    ha2 = a.forward_heads(x)
    for ta, tb in zip(ha, ha2):
        for u, v in zip(ta, tb):
            assert torch.equal(u, v)
    a.close()
    b.close()


def test_detect_workspace_survives_a_smaller_batch(pkg, dev, weights_reduced):
    """ron_detect's context-owned post-processing workspace is self-cleaning (its kernels leave the candidate counters zero, no memset
    per call): a call with a smaller batch in between must not disturb the counters of a later full batch.  (Laid out for the batch of
    the call, a batch of 1 once wrote its candidate keys over the counters of the next batch of 32.)"""
    x = torch.from_numpy(pkg['W'].synthetic_images(4, seed=9)).to(dev)
    net = pkg['ron'].RONNet(variant='reducedfc', dtype='bf16', max_batch=4, fuse_pools=True).load_weights(weights_reduced)
    a = net.detect(x)
    first = {k: getattr(a, k).clone() for k in ('count', 'classes', 'scores', 'bboxes', 'anchor_index')}
    net.detect(x[2:3])                       # other launch plan per convolution at batch 1: its own counts, not compared here
    for _ in range(2):
        b = net.detect(x)
        for k, v in first.items():
            assert torch.equal(getattr(b, k), v), k
    net.close()


@pytest.mark.parametrize('variant', ['reducedfc', 'full'])
@pytest.mark.parametrize('max_batch,head_plan', [(1, None), (4, 'batch'), (12, None), (16, None), (16, 'level'), (24, None), (32, None)])
def test_launch_plans_respect_dependencies(pkg, dev, weights_reduced, weights_full, variant, max_batch, head_plan):
    """Every launch plan plan_groups can produce, read back from the library and checked against the graph's true dependencies: a launch
    only reads what EARLIER launches wrote, and the members of a grouped launch neither read nor write what another member writes
    (they run concurrently).  A plan-table edit that breaks this would still pass every numerical test most of the time."""
    w = weights_reduced if variant == 'reducedfc' else weights_full
    net = pkg['ron'].RONNet(variant=variant, dtype='bf16', max_batch=max_batch, fuse_pools=True, head_plan=head_plan).load_weights(w)
    plan = [n for n in net.launch_plan()[:-1]]
    net.close()
    from plan_util import check_launches, head_dependencies
    dep = head_dependencies()
    start = plan.index('conv5_1') if 'conv5_1' in plan else next(i for i, n in enumerate(plan) if 'conv5_1' in n)
    launches = []
    for n in plan[start:]:
        if n.startswith('('):
            launches[-1].append(n[1:-1])
        elif n.startswith('group['):
            launches.append([n[len('group['):n.rindex('+')]])
        else:
            launches.append([n])
    seen_ops = [m for l in launches for m in l]
    assert sorted(seen_ops) == sorted(dep), sorted(set(dep) ^ set(seen_ops))          # every op exactly once
    check_launches(launches, dep)


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_fused_stem_vs_oracle_at_full_size(pkg, dev, weights_reduced, images, dtype):
    """conv1_1 + conv1_2 + pool1 as ONE kernel (stem2_kernel, 320 x 320 input) against the ORACLE, not against the other GPU
    path: pool1 and, two fused-pool convolutions further, pool2 vs the oracle with operands rounded to the storage type."""
    rnd = {'bf16': orf.round_bf16, 'fp16': orf.round_f16}[dtype]
    tol = {'bf16': 0.02, 'fp16': 0.003}[dtype]
    col = {}
    orf.ron_forward(images[:1], weights_reduced, 'reducedfc', backend='torch', round_fn=rnd, collect=col)
    net = pkg['ron'].RONNet(variant='reducedfc', dtype=dtype, max_batch=2, fuse_pools=True).load_weights(weights_reduced)
    net.forward_heads(torch.from_numpy(images).to(dev))
    for name in ('pool1', 'pool2'):
        got = net.end_point(name, 2).cpu().numpy()[:1]
        assert got.shape == col[name].shape
        assert _rel_err(got, col[name]) < tol, name
    net.close()
