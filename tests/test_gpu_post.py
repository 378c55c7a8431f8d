"""GPU parity: post-processing kernels (through the C ABI) vs the oracle and the golden vectors.

Bar: class ids, anchor indices, counts and order bit-exact; scores bit-exact when both sides start
from the same probabilities, 1e-6 otherwise; box coordinates within 1e-5 (expf vs numpy exp)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import anchors as oanchors  # noqa: E402
from oracle import np_post, synth  # noqa: E402

BOX_TOL = 1e-5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from ron_tensorflow_amd import ops as _ops
    return _ops


@pytest.fixture(scope='module')
def anchors():
    return oanchors.anchors_all_layers()


@pytest.fixture(scope='module')
def anchors_dev(ops, anchors, dev):
    return ops.anchors_to_device(anchors, dev)


def _to_dev(lst, dev):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in lst]


def _assert_same_dets(got, ref, scores_exact=True):
    assert got['classes'].shape == ref['classes'].shape, (got['classes'].shape, ref['classes'].shape)
    assert np.array_equal(got['classes'], ref['classes'])
    if 'anchor_index' in ref:
        assert np.array_equal(got['anchor_index'], ref['anchor_index'])
    if scores_exact:
        assert np.array_equal(got['scores'], ref['scores'])
    else:
        np.testing.assert_allclose(got['scores'], ref['scores'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got['bboxes'].reshape(-1, 4), ref['bboxes'].reshape(-1, 4), rtol=0, atol=BOX_TOL)


def test_softmax_matches_oracle(ops, dev):
    rs = np.random.RandomState(3)
    x = (rs.randn(4000, 21) * 3).astype(np.float32)
    y = ops.softmax_last(torch.from_numpy(x).to(dev)).cpu().numpy()
    np.testing.assert_allclose(y, np_post.softmax_last(x), rtol=0, atol=1e-6)
    o = (rs.randn(2, 5, 5, 10, 2) * 2).astype(np.float32)
    p = ops.softmax_last(torch.from_numpy(o).to(dev), pick=1).cpu().numpy()
    assert p.shape == (2, 5, 5, 10, 1)
    np.testing.assert_allclose(p, np_post.objectness_from_logits(o), rtol=0, atol=1e-6)


def test_decode_golden_and_oracle(ops, dev, anchors, anchors_dev, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g2_decode.npz'))
    for i in range(4):
        out = ops.bboxes_decode_layer(torch.from_numpy(g['loc%d' % i]).to(dev), anchors_dev[i]).cpu().numpy()
        np.testing.assert_allclose(out, g['dec%d' % i], rtol=0, atol=BOX_TOL)
    loc = np.random.RandomState(9).randn(3, 10, 10, 10, 4).astype(np.float32)     # batch > 1
    out = ops.bboxes_decode_layer(torch.from_numpy(loc).to(dev), anchors_dev[1]).cpu().numpy()
    np.testing.assert_allclose(out, np_post.bboxes_decode_layer(loc, anchors[1]), rtol=0, atol=BOX_TOL)


CASES = [  # seed, batch, bg, ob, cls_scale, select_thr, nms_thr
    (100, 1, 8.0, -4.0, 1.0, 0.01, 0.45),
    (101, 3, 8.0, -4.0, 1.0, 0.01, 0.45),
    (102, 2, 7.0, -3.0, 1.0, 0.01, 0.40),      # ~10 k candidates: radix-select path
    (103, 1, 4.0, -2.0, 1.0, 0.01, 0.45),      # ~190 k candidates
    (104, 2, 8.0, -2.0, 3.0, 0.5, 0.45),
    (105, 1, 30.0, -30.0, 1.0, 0.01, 0.45),    # nothing selected
    (106, 2, 2.0, 2.0, 0.05, 0.02, 0.30),      # flat scores ~1/21: heavy suppression, many near-equal scores
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'seed%d' % c[0])
def test_post_np_from_probabilities(ops, dev, anchors, anchors_dev, case):
    """GPU post-processing and oracle consume the SAME probabilities (made by the GPU softmax kernel)."""
    seed, batch, bg, ob, scale, thr, nms = case
    cls, obj, loc = synth.head_tensors(seed, batch=batch, bg=bg, ob=ob, cls_scale=scale)
    cls_d, obj_d, loc_d = _to_dev(cls, dev), _to_dev(obj, dev), _to_dev(loc, dev)
    pred_d = [ops.softmax_last(c) for c in cls_d]
    objp_d = [ops.softmax_last(o, pick=1) for o in obj_d]
    ref = np_post.detect_from_predictions([p.cpu().numpy() for p in pred_d], loc, anchors,
                                          objness_pred=[o.cpu().numpy() for o in objp_d], objectness_thres=0.03,
                                          select_threshold=thr, top_k=400, nms_threshold=nms)
    out, srt, ncand = ops.post_np(pred_d, objp_d, loc_d, anchors_dev, select_threshold=thr, nms_threshold=nms,
                                  cls_is_prob=True, obj_is_prob=True, want_sorted=True)
    got = out.to_lists()
    ncand = ncand.cpu().numpy()
    n_sorted = srt.count.cpu().numpy()
    for b in range(batch):
        assert ncand[b] == ref[b]['n_candidates']
        assert n_sorted[b] == ref[b]['n_sorted']
        _assert_same_dets(got[b], ref[b], scores_exact=True)
    # the fused path (softmax + gate inside the select kernel) must give the identical result
    out2, _, ncand2 = ops.post_np(cls_d, obj_d, loc_d, anchors_dev, select_threshold=thr, nms_threshold=nms)
    got2 = out2.to_lists()
    assert np.array_equal(ncand2.cpu().numpy(), ncand)
    for b in range(batch):
        for k in ('classes', 'scores', 'bboxes', 'anchor_index'):
            assert np.array_equal(got2[b][k], got[b][k]), (b, k)
    # records are zero padded behind `count`
    cnt = out.count.cpu().numpy()
    sc = out.scores.cpu().numpy()
    for b in range(batch):
        assert not sc[b, cnt[b]:].any()


def test_post_np_decoded_boxes_input(ops, dev, anchors, anchors_dev):
    """RONNet.bboxes_decode output fed back in (RON_IN_LOC_DECODED) == decode inside."""
    cls, obj, loc = synth.head_tensors(107, batch=2)
    cls_d, obj_d, loc_d = _to_dev(cls, dev), _to_dev(obj, dev), _to_dev(loc, dev)
    dec_d = [ops.bboxes_decode_layer(l, a) for l, a in zip(loc_d, anchors_dev)]
    a, _, _ = ops.post_np(cls_d, obj_d, loc_d, anchors_dev)
    b, _, _ = ops.post_np(cls_d, obj_d, dec_d, None, loc_decoded=True)
    for x, y in zip(a.to_lists(), b.to_lists()):
        for k in ('classes', 'scores', 'bboxes', 'anchor_index'):
            assert np.array_equal(x[k], y[k])


def test_post_np_golden_pipeline(ops, dev, anchors_dev, golden_dir):
    """Against the outputs the reference's np_methods produced (tests/golden/g3_pipeline.npz).
    Scores come from numpy's softmax there and from the device expf here: 1e-6, not bitwise."""
    g = np.load(os.path.join(golden_dir, 'g3_pipeline.npz'))
    for name in [str(n) for n in g['names']]:
        seed, bg, ob, scale, thr, nms = g[name + '/params']
        cls, obj, loc = synth.head_tensors(int(seed), batch=1, bg=bg, ob=ob, cls_scale=scale)
        out, srt, ncand = ops.post_np(_to_dev(cls, dev), _to_dev(obj, dev), _to_dev(loc, dev), anchors_dev,
                                      select_threshold=float(thr), nms_threshold=float(nms), want_sorted=True)
        got = out.to_lists()[0]
        assert int(ncand.cpu().numpy()[0]) == int(g[name + '/n_cand']), name
        ref = dict(classes=g[name + '/classes'], scores=g[name + '/scores'], bboxes=g[name + '/bboxes'])
        _assert_same_dets(got, ref, scores_exact=False)


def _run_list(ops, dev, classes, scores, boxes, top_k=400, thr=0.45):
    out, srt = ops.np_sort_nms(torch.from_numpy(classes.astype(np.int32))[None].to(dev),
                               torch.from_numpy(scores)[None].to(dev), torch.from_numpy(boxes)[None].to(dev),
                               top_k=top_k, nms_threshold=thr, want_sorted=True)
    return out.to_lists()[0], srt.to_lists()[0]


@pytest.mark.parametrize('case', ['ties', 'zero', 'inv', 'thr'])
def test_g4_edge_cases(ops, dev, golden_dir, case):
    g = np.load(os.path.join(golden_dir, 'g4_edge.npz'))
    c, s, b = g[case + '/in_classes'], g[case + '/in_scores'], g[case + '/in_bboxes']
    got, srt = _run_list(ops, dev, c, s, b)
    if case == 'ties':
        # order inside a tie is position-ascending (the oracle's definition; the reference's is unspecified)
        assert list(srt['anchor_index']) == [1, 4, 7, 0, 2, 3, 5, 6]
    assert np.array_equal(got['classes'], g[case + '/nms_classes'])
    assert np.array_equal(got['scores'], g[case + '/nms_scores'])
    assert np.array_equal(got['bboxes'], g[case + '/nms_bboxes'])


def test_g4_select_threshold_and_topk_cut(ops, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_edge.npz'))
    # stored as [1,1,1,500,21] / [1,1,1,500,4] (decoded boxes); same flat order as 50 cells x 10 anchors
    pred, boxes = g['cut/pred'].reshape(1, 50, 1, 10, 21), g['cut/boxes'].reshape(1, 50, 1, 10, 4)
    out, srt, ncand = ops.post_np([torch.from_numpy(pred).to(dev)], None, [torch.from_numpy(boxes).to(dev)], None,
                                  select_threshold=0.01, nms_threshold=2.0, cls_is_prob=True, loc_decoded=True,
                                  bbox_img=(-10., -10., 10., 10.), want_sorted=True)
    assert int(ncand.cpu().numpy()[0]) == 402
    s = srt.to_lists()[0]
    assert s['classes'].shape[0] == 400
    assert np.array_equal(s['classes'], g['cut/sorted_classes'])
    assert np.array_equal(s['scores'], g['cut/sorted_scores'])
    assert np.array_equal(s['bboxes'], g['cut/sorted_bboxes'])


def test_g4_clip_and_resize_against_a_non_unit_reference_box(ops, dev, golden_dir):
    """bboxes_clip and bboxes_resize with bbox_img != [0,0,1,1] (/root/reference/nets/np_methods.py:153-183) THROUGH
    ron_post_np: the five boxes of golden G4 (outside, straddling, inverted after the clip -- the np version has no
    repair) go in as decoded boxes of five different classes, so nothing is suppressed and the score order is the input
    order.  Sorted list = after the clip, final list = after the resize; both against the reference's own outputs."""
    g = np.load(os.path.join(golden_dir, 'g4_edge.npz'))
    boxes, ref2 = g['clip/in_bboxes'], g['clip/ref2']
    n = boxes.shape[0]
    pred = np.zeros((1, 1, 1, n, 21), np.float32)
    for k in range(n):
        pred[0, 0, 0, k, k + 1] = np.float32(0.9 - 0.1 * k)
    out, srt, ncand = ops.post_np([torch.from_numpy(pred).to(dev)], None, [torch.from_numpy(boxes.reshape(1, 1, 1, n, 4)).to(dev)],
                                  None, select_threshold=0.01, nms_threshold=0.45, cls_is_prob=True, loc_decoded=True,
                                  bbox_img=tuple(float(v) for v in ref2), want_sorted=True)
    assert int(ncand.cpu().numpy()[0]) == n
    s, o = srt.to_lists()[0], out.to_lists()[0]
    assert list(s['classes']) == list(range(1, n + 1)) and list(o['classes']) == list(range(1, n + 1))
    assert np.array_equal(s['bboxes'], g['clip/out_bboxes2'])                      # min / max: exact
    np.testing.assert_allclose(o['bboxes'], g['resize/out_bboxes2'], rtol=0, atol=1e-6)
    # and the unit box: clip only, resize is the identity
    out, srt, _ = ops.post_np([torch.from_numpy(pred).to(dev)], None, [torch.from_numpy(boxes.reshape(1, 1, 1, n, 4)).to(dev)],
                              None, select_threshold=0.01, nms_threshold=0.45, cls_is_prob=True, loc_decoded=True, want_sorted=True)
    assert np.array_equal(srt.to_lists()[0]['bboxes'], g['clip/out_bboxes'])
    assert np.array_equal(out.to_lists()[0]['bboxes'], g['clip/out_bboxes'])


def test_post_np_random_reference_boxes_vs_oracle(ops, dev, anchors, anchors_dev):
    """Whole pipeline with non-identity bbox_img (a crop of the image, and a box larger than it) vs the oracle."""
    cls, obj, loc = synth.head_tensors(108, batch=2)
    for ref_box in ((0.1, 0.2, 0.7, 0.9), (-0.25, -0.5, 1.5, 1.25)):
        ref = np_post.detect_from_logits(cls, obj, loc, anchors, bbox_img=ref_box)
        out, _, _ = ops.post_np(_to_dev(cls, dev), _to_dev(obj, dev), _to_dev(loc, dev), anchors_dev, bbox_img=ref_box)
        for got, r in zip(out.to_lists(), ref):
            _assert_same_dets(got, r, scores_exact=False)


def test_post_np_golden_pipeline_ssd512(ops, dev, golden_dir):
    """ron_post_np on 24 564 SSD-512 anchors (7 scales, 4 / 6 anchors per cell, no objectness) against what the
    reference's np_methods produced (tests/golden/g5_pipeline_ssd512.npz; SURVEY.md 8c "G5")."""
    from oracle import ssd_forward as osf
    adev = ops.anchors_to_device(osf.anchors_all_layers(), dev)
    g = np.load(os.path.join(golden_dir, 'g5_pipeline_ssd512.npz'))
    for name in [str(n) for n in g['names']]:
        seed, bg, scale, thr, nms = g[name + '/params']
        cls, loc = synth.ssd_head_tensors(int(seed), batch=1, bg=bg, cls_scale=scale)
        # the probabilities the reference pipeline was fed (numpy softmax): with 255 k candidates the 400 best scores are
        # ulps apart, and a device-side softmax that differs in the last bit would legitimately reorder them
        pred = [np_post.softmax_last(c) for c in cls]
        out, srt, ncand = ops.post_np(_to_dev(pred, dev), None, _to_dev(loc, dev), adev, select_threshold=float(thr),
                                      nms_threshold=float(nms), cls_is_prob=True, want_sorted=True)
        assert int(ncand.cpu().numpy()[0]) == int(g[name + '/n_cand']), name
        assert int(srt.count.cpu().numpy()[0]) == int(g[name + '/n_sorted']), name
        ref = dict(classes=g[name + '/classes'], scores=g[name + '/scores'], bboxes=g[name + '/bboxes'])
        _assert_same_dets(out.to_lists()[0], ref, scores_exact=True)
        s = srt.to_lists()[0]
        assert np.array_equal(s['classes'], g[name + '/sorted_classes']) and np.array_equal(s['scores'], g[name + '/sorted_scores'])


def test_list_sort_nms_random_vs_oracle(ops, dev):
    rs = np.random.RandomState(77)
    for n_in in (1, 63, 400, 401, 5000):
        classes = rs.randint(1, 4, n_in).astype(np.int64)
        scores = rs.permutation(n_in).astype(np.float32) / np.float32(n_in + 1) + np.float32(0.001)
        ctr = rs.uniform(.2, .8, (n_in, 2)).astype(np.float32)
        half = rs.uniform(.02, .2, (n_in, 2)).astype(np.float32)
        boxes = np.concatenate([ctr - half, ctr + half], axis=1).astype(np.float32)
        got, _ = _run_list(ops, dev, classes, scores, boxes)
        c, s, b, idx = np_post.bboxes_sort(classes, scores, boxes, top_k=400, extra=np.arange(n_in))
        c, s, b, idx = np_post.bboxes_nms(c, s, b, 0.45, extra=idx)
        assert np.array_equal(got['classes'], c)
        assert np.array_equal(got['anchor_index'], idx)
        assert np.array_equal(got['scores'], s)
        assert np.array_equal(got['bboxes'], b)


def test_list_sort_handles_negative_zero_and_nan_scores(ops, dev):
    """Caller-supplied lists may hold any float (raw logits): order of np.argsort(-scores), np_methods.py:137-150."""
    scores = np.array([0.5, -0.25, 0.0, -0.0, -3.0, 2.0, np.nan, -1e-30, 1e-30, -np.inf, np.inf], np.float32)
    n = scores.shape[0]
    classes = np.arange(1, n + 1).astype(np.int64)                  # all different: nothing is suppressed
    boxes = np.tile(np.array([[.1, .1, .5, .5]], np.float32), (n, 1))
    got, srt = _run_list(ops, dev, classes, scores, boxes)
    with np.errstate(invalid='ignore'):
        order = np.argsort(-scores, kind='stable')
    assert list(srt['anchor_index']) == list(order)
    assert list(got['anchor_index']) == list(order)


def _chain_boxes(n, w=0.002, d=0.0006, jitter=None):
    """Boxes i and i+1 overlap beyond 0.45 (IoU (w-d)/(w+d) = 0.54), i and i+2 do not (0.25): greedy NMS keeps every other box,
    and whether box i survives depends on box i-1, which depends on box i-2, ... -- the longest possible dependency chain."""
    x0 = np.arange(n, dtype=np.float32) * np.float32(d) + np.float32(0.01)
    if jitter is not None:
        x0 = x0 + jitter.astype(np.float32)
    b = np.stack([np.full(n, 0.1, np.float32), x0, np.full(n, 0.9, np.float32), x0 + np.float32(w)], axis=1)
    return np.ascontiguousarray(b.astype(np.float32))


@pytest.mark.parametrize('layout', ['one_class', 'two_classes', 'three_classes_mixed', 'class_id_70', 'jitter'])
def test_nms_long_suppression_chains(ops, dev, layout):
    """The class-wise scan decides 64 rows per fixed-point iteration and the generic scan likewise: chains of suppressions as long as
    a class, across the 64-row blocks, against np_methods' sequential loop (nets/np_methods.py:229-242)."""
    from oracle import np_post
    n = 400
    rs = np.random.RandomState(11)
    scores = np.linspace(0.99, 0.05, n).astype(np.float32)
    boxes = _chain_boxes(n, jitter=rs.uniform(-2e-4, 2e-4, n) if layout == 'jitter' else None)
    classes = {'one_class': np.full(n, 5), 'two_classes': 3 + 4 * (np.arange(n) % 2),
               'three_classes_mixed': np.array([1, 1, 2, 1, 3, 3, 2])[np.arange(n) % 7], 'class_id_70': np.full(n, 70),
               'jitter': 1 + (np.arange(n) // 150)}[layout].astype(np.int64)
    got, _ = _run_list(ops, dev, classes, scores, boxes)
    sc, ss, sb = np_post.bboxes_sort(classes, scores, boxes, top_k=400)
    rc, rsc, rb = np_post.bboxes_nms(sc, ss, sb, nms_threshold=0.45)
    assert np.array_equal(got['classes'], rc) and np.array_equal(got['scores'], rsc) and np.array_equal(got['bboxes'], rb)
    if layout in ('one_class', 'class_id_70'):
        assert len(rc) == n // 2                              # every other box
    if layout == 'two_classes':
        assert len(rc) == n                                   # same-class neighbours are two steps apart: nothing goes
