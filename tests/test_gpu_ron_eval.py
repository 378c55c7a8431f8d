"""GPU parity: the ron_eval.py post-processing (ron_post_eval) vs oracle/ron_eval_post.py -- labels, anchor indices, counts,
scores and boxes bit-exact when the same probabilities / decoded boxes go in."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import anchors as oanchors  # noqa: E402
from oracle import ron_eval_post as rp  # noqa: E402
from oracle import synth  # noqa: E402


def _to_dev(lst, dev):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in lst]


CASES = [  # seed, batch, bg, ob, obj_thr, sel_thr, nms_thr, keep_top_k, mode
    (300, 2, 2.0, 1.0, 0.5, 0.3, 0.4, 20, 'union'),        # a few hundred candidates, cut at keep_top_k
    (301, 1, 1.0, 2.0, 0.7, 0.2, 0.4, 200, 'union'),       # more candidates than the 512-row NMS window
    (302, 2, 3.0, 0.0, 0.5, 0.4, 0.3, 50, 'min'),
    (303, 1, 30.0, -30.0, 0.95, 0.6, 0.4, 20, 'union'),    # nothing passes
    # 20 167 / 16 056 candidates and heavy suppression: kept rows sit at score ranks up to 19 174, far beyond one
    # 1 024-row pass of the NMS kernel (the reference, ron_eval.py:111-144, takes all candidates)
    (304, 1, 0.0, 3.0, 0.1, 0.05, 0.1, 400, 'union'),
    (305, 1, 1.0, 2.0, 0.3, 0.1, 0.05, 300, 'min'),
]


@pytest.mark.parametrize('by_class', [False, True, 'scores'], ids=['agnostic', 'by_class', 'by_class_scores'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'seed%d' % c[0])
def test_post_eval_matches_oracle(case, by_class):
    """by_class True: tf_bboxes_nms_by_class_v1 (ron_eval.py:282-366) instead of tf_bboxes_nms (:146-206); 'scores': tf_bboxes_nms_by_class
    (:212-280), one NMS per score column, rows back in anchor order."""
    from ron_tensorflow_amd import ops, ron_eval
    seed, batch, bg, ob, obj_thr, sel_thr, nms_thr, keep, mode = case
    dev = torch.device('cuda:0')
    anchors = oanchors.anchors_all_layers()
    adev = ops.anchors_to_device(anchors, dev)
    cls, obj, loc = synth.head_tensors(seed, batch=batch, bg=bg, ob=ob)
    cls = [c * np.float32(3.0) for c in cls]                 # sharper class distributions: some products exceed the threshold
    loc = [l * np.float32(0.2) for l in loc]
    cls_d, obj_d, loc_d = _to_dev(cls, dev), _to_dev(obj, dev), _to_dev(loc, dev)
    pred = [ops.softmax_last(c) for c in cls_d]
    objp = [ops.softmax_last(o, pick=1) for o in obj_d]
    dec = [ops.bboxes_decode_layer(l, a) for l, a in zip(loc_d, adev)]
    shapes = [(375, 500), (500, 333)][:batch]
    det = ron_eval.post_eval(pred, objp, dec, None, shapes, objectness_thres=obj_thr, select_threshold=sel_thr,
                             nms_threshold=nms_thr, keep_top_k=keep, nms_mode=mode, nms_by_class=by_class)
    got = det.to_lists()
    n_total = 0
    for i in range(batch):
        ref = rp.post_eval_image([p[i].cpu().numpy() for p in pred], [o[i].cpu().numpy() for o in objp],
                                 [d[i].cpu().numpy() for d in dec], shapes[i], objectness_thres=obj_thr,
                                 select_threshold=sel_thr, nms_threshold=nms_thr, keep_top_k=keep, nms_mode=mode,
                                 nms_by_class=by_class)
        g = got[i]
        assert np.array_equal(g['classes'], ref['classes']), i
        assert np.array_equal(g['anchor_index'], ref['anchor_index']), i
        assert np.array_equal(g['scores'], ref['scores']) and np.array_equal(g['bboxes'], ref['bboxes'])
        n_total += len(ref['classes'])
    if seed != 303:
        assert n_total > 0
    else:
        assert n_total == 0 and int(det.count.sum()) == 0
    # fused entry: logits, objectness logits, raw offsets
    det2 = ron_eval.post_eval(cls_d, obj_d, loc_d, adev, shapes, objectness_thres=obj_thr, select_threshold=sel_thr,
                              nms_threshold=nms_thr, keep_top_k=keep, nms_mode=mode, cls_is_prob=False, obj_is_prob=False,
                              loc_decoded=False, nms_by_class=by_class)
    assert torch.equal(det2.count, det.count) and torch.equal(det2.classes, det.classes)
    assert torch.equal(det2.anchor_index, det.anchor_index)
    assert float((det2.bboxes - det.bboxes).abs().max()) <= 1e-6


def test_unknown_mode_raises():
    from ron_tensorflow_amd import ron_eval
    with pytest.raises(ValueError):
        ron_eval.post_eval([torch.zeros((1, 1, 1, 1, 21), device='cuda')], None, None, None, [(1, 1)], nms_mode='iou')


def test_by_class_scores_hand_case_on_device():
    """The hand case of tests/test_oracle_ron_eval.py::test_by_class_scores_variant_hand_case through ron_post_eval (nms_mode | 4): a row
    relabelled by the column that kept it, a row kept by the background column, rows in anchor order."""
    from ron_tensorflow_amd import ron_eval
    dev = torch.device('cuda:0')
    pred = np.array([[.10, .50, .40], [.10, .45, .45], [.05, .15, .80], [.40, .45, .15], [.30, .55, .15]], np.float32).reshape(1, 1, 1, 5, 3)
    obj = np.array([.99, .98, .97, .96, .995], np.float32).reshape(1, 1, 1, 5, 1)
    box = np.array([[.1, .1, .5, .5], [.12, .1, .5, .5], [.6, .6, .9, .9], [.1, .6, .4, .9], [.11, .6, .4, .9]], np.float32).reshape(1, 1, 1, 5, 4)
    det = ron_eval.post_eval(_to_dev([pred], dev), _to_dev([obj], dev), _to_dev([box], dev), None, [(320, 320)], num_classes=3,
                             objectness_thres=0.95, select_threshold=0.3, nms_threshold=0.4, keep_top_k=20, nms_by_class='scores')
    g = det.to_lists()[0]
    ref = rp.post_eval_image([pred[0]], [obj[0]], [box[0]], (320, 320), objectness_thres=0.95, select_threshold=0.3, nms_threshold=0.4,
                             keep_top_k=20, nms_by_class='scores')
    assert g['anchor_index'].tolist() == [0, 1, 2, 3, 4] and g['classes'].tolist() == [1, 2, 2, 0, 1]
    assert np.array_equal(g['scores'], ref['scores']) and np.array_equal(g['bboxes'], ref['bboxes'])
    assert det.capacity == 3 * 20
