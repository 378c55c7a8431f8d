"""GPU parity: TF-evaluation post-processing variant (ron_post_tfe) vs oracle/tfe_post.py: scores bit-exact
(same probabilities in), boxes bit-exact when decoded boxes are handed over."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import anchors as oanchors  # noqa: E402
from oracle import np_post, synth, tfe_post  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _to_dev(lst, dev):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in lst]


CASES = [  # seed, batch, bg, ob, select_thr, nms_thr, top_k, keep_top_k, mode, min_size
    (200, 2, 8.0, -4.0, 0.01, 0.4, 200, 100, 'min', 0.03),       # eval_ron_network.py flags
    (201, 1, 6.0, -2.0, 0.01, 0.4, 200, 100, 'union', 0.03),
    (202, 2, 4.0, -2.0, 0.01, 0.45, 400, 200, 'min', None),      # dense, no size filter (SSD call order)
    (203, 1, 30.0, -30.0, 0.01, 0.4, 200, 100, 'min', 0.03),     # empty
    (204, 1, 7.0, -3.0, 0.01, 0.3, 64, 8, 'min', 0.03),          # keep_top_k cuts the NMS loop
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'seed%d' % c[0])
def test_detected_bboxes_matches_oracle(dev, case):
    from ron_tensorflow_amd import ops, tfe
    seed, batch, bg, ob, thr, nms, top_k, keep, mode, min_size = case
    anchors = oanchors.anchors_all_layers()
    adev = ops.anchors_to_device(anchors, dev)
    cls, obj, loc = synth.head_tensors(seed, batch=batch, bg=bg, ob=ob)
    # make overlap likely: shrink the offsets so neighbouring anchors give nearly the same box
    loc = [l * np.float32(0.2) for l in loc]
    cls_d, obj_d, loc_d = _to_dev(cls, dev), _to_dev(obj, dev), _to_dev(loc, dev)
    pred = [ops.softmax_last(c) for c in cls_d]
    objp = [ops.softmax_last(o, pick=1) for o in obj_d]
    # eval_ron_network.py:226-229: decode, then gate the predictions by objectness
    dec = [ops.bboxes_decode_layer(l, a) for l, a in zip(loc_d, adev)]
    gated = [(o > 0.03).to(torch.float32) * p for o, p in zip(objp, pred)]
    ds, db = tfe.detected_bboxes(gated, dec, num_classes=21, select_threshold=thr, nms_threshold=nms,
                                 clipping_bbox=[0., 0., 1., 1.], top_k=top_k, keep_top_k=keep, nms_mode=mode,
                                 min_size=min_size)
    rs, rb = tfe_post.detected_bboxes([g.cpu().numpy() for g in gated], [d.cpu().numpy() for d in dec], num_classes=21,
                                      select_threshold=thr, nms_threshold=nms, clipping_bbox=[0., 0., 1., 1.],
                                      top_k=top_k, keep_top_k=keep, nms_mode=mode, min_size=min_size)
    n_kept = 0
    for c in range(1, 21):
        assert tuple(ds[c].shape) == (batch, keep) and tuple(db[c].shape) == (batch, keep, 4)
        assert np.array_equal(ds[c].cpu().numpy(), rs[c]), c
        assert np.array_equal(db[c].cpu().numpy(), rb[c]), c
        n_kept += int((rs[c] > 0).sum())
    if seed not in (203,):
        assert n_kept > 0
    # fused entry (logits + objectness logits + raw offsets in one call) gives the same lists
    s2, b2 = tfe.post_tfe(cls_d, obj_d, loc_d, adev, select_threshold=thr, nms_threshold=nms,
                          clipping_bbox=[0., 0., 1., 1.], top_k=top_k, keep_top_k=keep, nms_mode=mode, min_size=min_size,
                          cls_is_prob=False, obj_is_prob=False, loc_decoded=False)
    for c in range(1, 21):
        assert np.array_equal(s2[:, c - 1].cpu().numpy(), rs[c])
        np.testing.assert_allclose(b2[:, c - 1].cpu().numpy(), rb[c], rtol=0, atol=1e-6)


def test_unknown_nms_mode_raises(dev):
    from ron_tensorflow_amd import tfe
    with pytest.raises(ValueError):
        tfe.post_tfe([torch.zeros((1, 1, 1, 1, 21), device=dev)], None, [torch.zeros((1, 1, 1, 1, 4), device=dev)], None,
                     nms_mode='iou')


@pytest.mark.parametrize('mode,keep', [('min', 200), ('union', 200), ('union', 37)])
def test_long_suppression_chain_per_class(dev, mode, keep):
    """One class holds a chain of boxes in which every box overlaps only its neighbours: the kept set alternates along the whole list,
    across the 64-row blocks of the scan, and keep_top_k cuts it in the middle (tf_extended/bboxes.py:173-234)."""
    from ron_tensorflow_amd import tfe
    shapes = [(5, 5), (10, 10), (20, 20), (40, 40)]
    pred = [np.zeros((1, h, w, 10, 21), np.float32) for h, w in shapes]
    boxes = [np.zeros((1, h, w, 10, 4), np.float32) for h, w in shapes]
    n = 200
    x0 = (np.arange(n) * 0.0006 + 0.01).astype(np.float32)
    # 'min' overlap of neighbours: (w - d) / w = 0.7, of second neighbours 0.4; 'union': 0.54 / 0.25
    chain = np.stack([np.full(n, 0.1, np.float32), x0, np.full(n, 0.9, np.float32), x0 + np.float32(0.002)], axis=1)
    flat_p, flat_b = pred[3].reshape(-1, 21), boxes[3].reshape(-1, 4)
    flat_p[:n, 7] = np.linspace(0.95, 0.2, n).astype(np.float32)
    flat_b[:n] = chain
    flat_p[n:2 * n:3, 2] = 0.5                      # a second class with identical boxes: one survivor
    flat_b[n:2 * n:3] = np.array([0.2, 0.2, 0.6, 0.6], np.float32)
    thr = 0.45
    ds, db = tfe.detected_bboxes(_to_dev(pred, dev), _to_dev(boxes, dev), num_classes=21, select_threshold=0.01, nms_threshold=thr,
                                 clipping_bbox=[0., 0., 1., 1.], top_k=200, keep_top_k=keep, nms_mode=mode, min_size=None)
    rs, rb = tfe_post.detected_bboxes(pred, boxes, num_classes=21, select_threshold=0.01, nms_threshold=thr,
                                      clipping_bbox=[0., 0., 1., 1.], top_k=200, keep_top_k=keep, nms_mode=mode, min_size=None)
    for c in range(1, 21):
        assert np.array_equal(ds[c].cpu().numpy(), rs[c]), c
        assert np.array_equal(db[c].cpu().numpy(), rb[c]), c
    assert int((rs[7] > 0).sum()) == min(keep, n // 2) and int((rs[2] > 0).sum()) == 1
