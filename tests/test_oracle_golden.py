"""Pin the oracle (oracle/anchors.py, oracle/np_post.py) against golden vectors that
tests/golden/make_golden.py produced by running the reference's own numpy code
(nets/np_methods.py, nets/ron_vgg_320.py anchor functions).  CPU only."""
import os

import numpy as np
import pytest

from oracle import anchors as oanchors
from oracle import np_post, synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_g1_anchors_bit_exact(golden_dir):
    g = _load(golden_dir, 'g1_anchors_ron320.npz')
    layers = oanchors.anchors_all_layers()
    assert len(layers) == 4
    for i, (y, x, h, w) in enumerate(layers):
        for nm, arr in (('y', y), ('x', x), ('h', h), ('w', w)):
            ref = g['%s%d' % (nm, i)]
            assert arr.dtype == np.float32 and arr.shape == ref.shape
            assert np.array_equal(arr, ref), (nm, i)
    # spot values quoted in SURVEY.md §8(a2)
    assert layers[0][0][0, 0, 0] == np.float32(0.1)
    np.testing.assert_allclose(layers[0][2][:4], [.7, .8, .49497475, .56568545], rtol=1e-6)


def test_g2_decode_bit_exact(golden_dir):
    g = _load(golden_dir, 'g2_decode.npz')
    layers = oanchors.anchors_all_layers()
    for i, a in enumerate(layers):
        got = np_post.bboxes_decode_layer(g['loc%d' % i], a)
        assert np.array_equal(got, g['dec%d' % i]), i


def test_decode_is_batch_capable():
    layers = oanchors.anchors_all_layers()
    rs = np.random.RandomState(5)
    loc = rs.randn(3, 5, 5, 10, 4).astype(np.float32)
    full = np_post.bboxes_decode_layer(loc, layers[0])
    for b in range(3):
        assert np.array_equal(full[b:b + 1], np_post.bboxes_decode_layer(loc[b:b + 1], layers[0]))


def assert_equal_up_to_tie_order(got, ref):
    """(classes, scores, bboxes) equal row by row, except that rows inside a run of exactly
    equal scores may be permuted: np.argsort in the reference is unstable (SIMD introsort), so
    its order inside such a run is unspecified; the oracle defines it as position-ascending."""
    gc, gs, gb = got
    rc, rs_, rb = ref
    assert np.array_equal(gs, rs_)
    start = 0
    n = len(gs)
    while start < n:
        end = start
        while end < n and gs[end] == gs[start]:
            end += 1
        rows_g = sorted((int(gc[i]),) + tuple(float(v) for v in gb[i]) for i in range(start, end))
        rows_r = sorted((int(rc[i]),) + tuple(float(v) for v in rb[i]) for i in range(start, end))
        assert rows_g == rows_r
        start = end


def _g3_names(g):
    return [str(n) for n in g['names']]


def test_g3_pipeline_matches_reference(golden_dir):
    g = _load(golden_dir, 'g3_pipeline.npz')
    layers = oanchors.anchors_all_layers()
    for name in _g3_names(g):
        seed, bg, ob, scale, thr, nms = g[name + '/params']
        cls, obj, loc = synth.head_tensors(int(seed), batch=1, bg=bg, ob=ob, cls_scale=scale)
        res = np_post.detect_from_logits(cls, obj, loc, layers, objectness_thres=0.03,
                                         select_threshold=thr, top_k=400, nms_threshold=nms)[0]
        assert res['n_candidates'] == int(g[name + '/n_cand']), name
        assert res['n_sorted'] == int(g[name + '/n_sorted']), name
        assert np.array_equal(res['classes'], g[name + '/classes']), name
        assert np.array_equal(res['scores'], g[name + '/scores']), name
        assert np.array_equal(res['bboxes'].reshape(-1, 4), g[name + '/bboxes']), name
        # anchor-index side channel is consistent with the kept boxes
        assert res['anchor_index'].shape == res['classes'].shape
        if res['anchor_index'].size:
            assert res['anchor_index'].min() >= 0 and res['anchor_index'].max() < 21250


def test_select_threshold_none_is_the_argmax_branch():
    """np_methods.py:82: `select_threshold is None or select_threshold == 0` -- both spellings, one candidate per anchor."""
    layers = oanchors.anchors_all_layers()
    cls, obj, loc = synth.head_tensors(10, batch=1, bg=4.5, ob=-2.0)
    a = np_post.detect_from_logits(cls, obj, loc, layers, select_threshold=None)[0]
    b = np_post.detect_from_logits(cls, obj, loc, layers, select_threshold=0)[0]
    assert a['n_candidates'] == b['n_candidates'] == 170
    for k in ('classes', 'scores', 'bboxes', 'anchor_index'):
        assert np.array_equal(a[k], b[k])
    assert len(set(a['anchor_index'].tolist())) == len(a['anchor_index'])       # at most one class per anchor
    assert a['classes'].min() >= 1


def test_g3_batch_equals_per_image(golden_dir):
    layers = oanchors.anchors_all_layers()
    cls, obj, loc = synth.head_tensors(40, batch=3)
    res = np_post.detect_from_logits(cls, obj, loc, layers)
    for b in range(3):
        one = np_post.detect_from_logits([c[b:b + 1] for c in cls], [o[b:b + 1] for o in obj],
                                         [l[b:b + 1] for l in loc], layers)[0]
        for k in ('classes', 'scores', 'bboxes', 'anchor_index'):
            assert np.array_equal(one[k], res[b][k])


@pytest.mark.parametrize('case', ['ties', 'zero', 'inv', 'thr'])
def test_g4_nms_cases(golden_dir, case):
    g = _load(golden_dir, 'g4_edge.npz')
    if case == 'ties':
        c, s, b = np_post.bboxes_sort(g['ties/in_classes'], g['ties/in_scores'], g['ties/in_bboxes'], top_k=400)
        assert_equal_up_to_tie_order((c, s, b), (g['ties/sorted_classes'], g['ties/sorted_scores'],
                                                 g['ties/sorted_bboxes']))
        # oracle's own definition: stable, i.e. position-ascending inside a tie
        assert list(np_post.sort_order(g['ties/in_scores'], 400)) == [1, 4, 7, 0, 2, 3, 5, 6]
    else:
        c, s, b = g[case + '/in_classes'], g[case + '/in_scores'], g[case + '/in_bboxes']
    c2, s2, b2 = np_post.bboxes_nms(c, s, b, nms_threshold=0.45)
    assert np.array_equal(c2, g[case + '/nms_classes'])
    assert np.array_equal(s2, g[case + '/nms_scores'])
    assert np.array_equal(b2, g[case + '/nms_bboxes'])


def test_g4_clip_resize(golden_dir):
    g = _load(golden_dir, 'g4_edge.npz')
    assert np.array_equal(np_post.bboxes_clip([0., 0., 1., 1.], g['clip/in_bboxes']), g['clip/out_bboxes'])
    out2 = np_post.bboxes_clip(g['clip/ref2'], g['clip/in_bboxes'])
    assert np.array_equal(out2, g['clip/out_bboxes2'])
    assert np.array_equal(np_post.bboxes_resize(g['clip/ref2'], out2), g['resize/out_bboxes2'])


def test_g4_select_threshold_and_topk_cut(golden_dir):
    g = _load(golden_dir, 'g4_edge.npz')
    pred, boxes = g['cut/pred'], g['cut/boxes']
    c, s, b, ai = np_post.bboxes_select_image([pred[0]], [boxes[0]], 0.01)
    assert c.shape[0] == 402                       # 401 + the one just above thr; the == thr one is out
    assert np.array_equal(c, g['cut/sel_classes'])
    assert np.array_equal(s, g['cut/sel_scores'])
    assert np.array_equal(b, g['cut/sel_bboxes'])
    c, s, b, ai = np_post.bboxes_sort(c, s, b, top_k=400, extra=ai)
    assert c.shape[0] == 400
    assert np.array_equal(c, g['cut/sorted_classes'])
    assert np.array_equal(s, g['cut/sorted_scores'])
    assert np.array_equal(b, g['cut/sorted_bboxes'])


def test_g5_ssd512_anchors_bit_exact(golden_dir):
    from oracle import ssd_forward as osf
    g = _load(golden_dir, 'g5_anchors_ssd512.npz')
    layers = osf.anchors_all_layers()
    assert [len(l[2]) for l in layers] == [4, 6, 6, 6, 6, 4, 4]
    assert sum(l[0].shape[0] * l[0].shape[1] * len(l[2]) for l in layers) == 24564
    for i, (y, x, h, w) in enumerate(layers):
        for nm, arr in (('y', y), ('x', x), ('h', h), ('w', w)):
            assert np.array_equal(arr, g['%s%d' % (nm, i)]), (nm, i)


def test_g5_ssd512_pipeline_matches_reference(golden_dir):
    """np_methods pipeline on the 24 564 SSD-512 anchors (no objectness gate), outputs of the reference's own numpy code
    (/root/reference/nets/np_methods.py:100-131, 137-150, 153-183, 229-242; SURVEY.md 8c "G5")."""
    from oracle import ssd_forward as osf
    g = _load(golden_dir, 'g5_pipeline_ssd512.npz')
    layers = osf.anchors_all_layers()
    for name in [str(n) for n in g['names']]:
        seed, bg, scale, thr, nms = g[name + '/params']
        cls, loc = synth.ssd_head_tensors(int(seed), batch=1, bg=bg, cls_scale=scale)
        pred = [np_post.softmax_last(c) for c in cls]
        res = np_post.detect_from_predictions(pred, loc, layers, objness_pred=None, select_threshold=thr, top_k=400,
                                              nms_threshold=nms)[0]
        assert res['n_candidates'] == int(g[name + '/n_cand']), name
        assert res['n_sorted'] == int(g[name + '/n_sorted']), name
        assert np.array_equal(res['classes'], g[name + '/classes']), name
        assert np.array_equal(res['scores'], g[name + '/scores']), name
        assert np.array_equal(res['bboxes'].reshape(-1, 4), g[name + '/bboxes']), name
        if res['anchor_index'].size:
            assert res['anchor_index'].min() >= 0 and res['anchor_index'].max() < 24564
