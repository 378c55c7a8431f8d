"""CPU: evaluation bookkeeping oracle vs the reference's numpy voc_ap golden vectors, host metrics vs the oracle."""
import os

import numpy as np
import pytest

from oracle import eval_metrics as em
from ron_tensorflow_amd import metrics

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'g6_voc_ap.npz')


@pytest.mark.parametrize('k', range(6))
def test_ap_vs_reference_voc_ap(k):
    g = np.load(GOLD)
    prec, rec = g['prec%d' % k], g['rec%d' % k]
    for fn in (em.average_precision_voc07, metrics.average_precision_voc07):
        assert abs(fn(prec, rec) - float(g['ap07_%d' % k])) < 1e-12
    for fn in (em.average_precision_voc12, metrics.average_precision_voc12):
        assert abs(fn(prec, rec) - float(g['ap12_%d' % k])) < 1e-12


def test_matching_hand_case():
    # two ground truths of class 1 (one difficult), one of class 2
    gl = np.array([1, 1, 2, 0])
    gb = np.array([[.1, .1, .5, .5], [.6, .6, .9, .9], [.1, .1, .5, .5], [0, 0, 0, 0]], np.float32)
    gd = np.array([0, 1, 0, 0])
    det = np.array([[.1, .1, .5, .5],        # tp on gt 0
                    [.12, .1, .5, .5],       # duplicate -> fp
                    [.6, .6, .9, .9],        # difficult -> neither
                    [.7, .1, .9, .3],        # no overlap: argmax 0 (not difficult) -> fp
                    [0, 0, 0, 0]], np.float32)   # zero padding -> fp (removed later by the score filter)
    sc = np.array([.9, .8, .7, .6, 0.], np.float32)
    n, tp, fp = em.bboxes_matching(1, sc, det, gl, gb, gd)
    assert n == 1
    assert tp.tolist() == [True, False, False, False, False]
    assert fp.tolist() == [False, True, False, True, True]
    t, f, s = em.streaming_filter(tp, fp, sc)
    assert t.tolist() == [True, False, False] and f.tolist() == [False, True, True]
    prec, rec = em.precision_recall(n, t, f, s)
    assert np.allclose(prec, [1, .5, 1 / 3.]) and np.allclose(rec, [1, 1, 1])


def test_host_metrics_match_oracle():
    rs = np.random.RandomState(4)
    labels = [1, 2, 3]
    st = metrics.StreamingTpFp(labels)
    acc = {c: [0, [], [], []] for c in labels}
    for _ in range(3):
        n, k = 4, 50
        tp = rs.rand(n, 3, k) < 0.3
        fp = ~tp & (rs.rand(n, 3, k) < 0.8)
        sc = np.sort(rs.rand(n, 3, k).astype(np.float32), axis=-1)[..., ::-1].copy()
        sc[..., 40:] = 0
        ngb = rs.randint(0, 6, (n, 3))
        st.update(ngb, tp, fp, sc)
        for i, c in enumerate(labels):
            t, f, s = em.streaming_filter(tp[:, i], fp[:, i], sc[:, i])
            acc[c][0] += int(ngb[:, i].sum())
            acc[c][1].append(t); acc[c][2].append(f); acc[c][3].append(s)
    res = metrics.evaluate(st)
    aps = []
    for c in labels:
        t, f, s = (np.concatenate(x) for x in acc[c][1:])
        ngb, ndet, t2, f2, s2 = st.arrays(c)
        assert ngb == acc[c][0] and ndet == t.shape[0]
        assert np.array_equal(t, t2) and np.array_equal(f, f2) and np.array_equal(s, s2)
        prec, rec = em.precision_recall(acc[c][0], t, f, s)
        p2, r2 = metrics.precision_recall(ngb, ndet, t2, f2, s2)
        assert np.array_equal(prec, p2) and np.array_equal(rec, r2)
        assert res['AP_VOC07/%d' % c] == em.average_precision_voc07(prec, rec)
        assert res['AP_VOC12/%d' % c] == em.average_precision_voc12(prec, rec)
        aps.append(res['AP_VOC12/%d' % c])
    assert abs(res['AP_VOC12/mAP'] - np.mean(aps)) < 1e-15


def test_no_ground_truth_and_no_detections():
    prec, rec = em.precision_recall(0, np.array([False]), np.array([True]), np.array([.5], np.float32))
    assert prec.tolist() == [0.] and rec.tolist() == [0.]
    z = np.zeros((0,))
    assert em.average_precision_voc12(z, z) == 0.
    assert metrics.average_precision_voc07(z, z) == 0.


GOLD7 = os.path.join(os.path.dirname(__file__), 'golden', 'g7_voc_eval.npz')


@pytest.mark.parametrize('cls', range(3))
def test_voc_eval_vs_reference(cls):
    """Oracle and host metric vs what the reference's DetectorEvalPascal.voc_eval returned for the same detections / XML
    ground truth (tests/golden/make_golden.py::g7_voc_eval): recall and precision arrays and both APs, exactly."""
    g = np.load(GOLD7)
    gt = g['gt']                                     # (image, class, difficult, xmin, ymin, xmax, ymax) as in the XML
    gt_boxes, gt_diff = {}, {}
    for i in range(12):
        rows = gt[(gt[:, 0] == i) & (gt[:, 1] == cls)]
        gt_boxes[i] = (rows[:, 3:7] - 1).astype(np.float64)          # parse_rec subtracts one (voc_eval.py:67-70)
        gt_diff[i] = rows[:, 2].astype(bool)
    det = g['det_%d' % cls]
    for fn in (em.voc_eval_class, metrics.voc_eval_class):
        for use07, key in ((True, 'ap07_%d'), (False, 'ap12_%d')):
            rec, prec, ap = fn(det[:, 0].astype(int), g['score_%d' % cls], det[:, 1:5], gt_boxes, gt_diff, 0.5, use07)
            assert np.array_equal(rec, g['rec_%d' % cls]) and np.array_equal(prec, g['prec_%d' % cls])
            assert abs(ap - float(g[key % cls])) < 1e-15
    assert metrics.voc_eval_class([], [], np.zeros((0, 4)), gt_boxes, gt_diff) == (-1., -1., -1.)
