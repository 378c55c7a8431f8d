"""CPU: hand-derived cases for the TF-variant oracle (oracle/tfe_post.py).  The reference has no test for
this path and TensorFlow cannot run here, so these pin the restatement to the semantics read from
tf_extended/bboxes.py:173-234 and nets/ron_vgg_320.py:196-256 by hand."""
import numpy as np

from oracle import tfe_post

F = np.float32


def test_min_mode_suppresses_nested_box_union_mode_does_not():
    big = np.array([0., 0., 1., 1.], F)
    small = np.array([.4, .4, .6, .6], F)          # inside big: inter/min(area) = 1, IoU = 0.04
    scores = np.array([.9, .8], F)
    boxes = np.stack([big, small])
    s, b = tfe_post.nms_one_class(scores, boxes, 0.5, 4, 'min')
    assert list(s) == [F(.9), 0, 0, 0] and np.array_equal(b[0], big) and not b[1:].any()
    s, b = tfe_post.nms_one_class(scores, boxes, 0.5, 4, 'union')
    assert list(s[:2]) == [F(.9), F(.8)]


def test_keep_top_k_limits_iterations_and_zero_rows_are_inert():
    boxes = np.array([[.0, .0, .1, .1], [.2, .2, .3, .3], [.4, .4, .5, .5], [0, 0, 0, 0]], F)
    scores = np.array([.9, .8, .7, 0.], F)
    s, b = tfe_post.nms_one_class(scores, boxes, 0.5, 2, 'min')
    assert list(s) == [F(.9), F(.8)]
    s, b = tfe_post.nms_one_class(scores, boxes, 0.5, 6, 'min')
    assert list(s) == [F(.9), F(.8), F(.7), 0, 0, 0] and not b[3:].any()


def test_overlap_exactly_at_threshold_suppresses():
    a = np.array([0., 0., 1., 1.], F)
    b = np.array([0., 0., 1., .5], F)            # inter/min = .5/.5 = 1 ; union: .5/1 = .5 -> NOT (< .5) -> suppressed
    s, _ = tfe_post.nms_one_class(np.array([.9, .8], F), np.stack([a, b]), 0.5, 2, 'union')
    assert list(s) == [F(.9), 0]


def test_clip_repairs_inverted_boxes():
    out = tfe_post.clip_with_repair([0., 0., 1., 1.], np.array([[1.2, 1.3, 1.5, 1.6], [-.5, .2, .5, .8]], F))
    assert np.array_equal(out[0], np.array([1., 1., 1., 1.], F))      # ymin/xmin pulled back onto ymax/xmax
    assert np.array_equal(out[1], np.array([0., .2, .5, .8], F))


def test_detected_bboxes_pipeline_small():
    # one layer, 1x1 cell, 4 anchors, 3 classes
    pred = np.zeros((1, 1, 1, 4, 3), F)
    pred[0, 0, 0, :, 1] = [.6, .5, .005, .7]
    pred[0, 0, 0, :, 2] = [.0, .3, .9, .0]
    loc = np.array([[[[[.1, .1, .5, .5], [.12, .1, .5, .52], [.6, .6, .9, .9], [.2, .2, .21, .8]]]]], F)
    ds, db = tfe_post.detected_bboxes([pred], [loc], num_classes=3, select_threshold=.01, nms_threshold=.4,
                                      clipping_bbox=[0., 0., 1., 1.], top_k=4, keep_top_k=3, nms_mode='min')
    # class 1: anchor 3 (.7) fails the min-size filter (h = .01), anchor 2 is under thr, anchors 0/1 overlap -> keep .6
    assert list(ds[1][0]) == [F(.6), 0, 0]
    assert np.array_equal(db[1][0, 0], loc[0, 0, 0, 0])
    # class 2: anchors 2 (.9) and 1 (.3) are disjoint -> both kept, sorted
    assert list(ds[2][0]) == [F(.9), F(.3), 0]
    assert np.array_equal(db[2][0, 1], loc[0, 0, 0, 1])
