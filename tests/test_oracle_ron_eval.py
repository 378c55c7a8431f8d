"""CPU: hand case for the ron_eval.py post-processing oracle (TF graph code, parity unpinned)."""
import numpy as np

from oracle import ron_eval_post as rp


def test_hand_case():
    # one layer, 1 x 1 x 6 anchors, 3 classes
    pred = np.array([[.1, .8, .1],      # label 1, score .8 * obj
                     [.1, .7, .2],      # label 1, overlaps anchor 0 -> suppressed (class agnostic NMS would also drop a label-2 box)
                     [.2, .1, .7],      # label 2, far away
                     [.9, .05, .05],    # background
                     [.1, .2, .7],      # label 2 but objectness too low
                     [.1, .1, .8]],     # label 2, box too small
                    np.float32).reshape(1, 1, 6, 3)
    obj = np.array([.99, .98, .97, .99, .5, .99], np.float32).reshape(1, 1, 6, 1)
    box = np.array([[.1, .1, .5, .5], [.12, .1, .5, .5], [.6, .6, .9, .9], [.1, .1, .5, .5], [.6, .6, .9, .9],
                    [.3, .3, .31, .31]], np.float32).reshape(1, 1, 6, 4)
    r = rp.post_eval_image([pred], [obj], [box], (320, 320), objectness_thres=0.95, select_threshold=0.6, nms_threshold=0.4,
                           keep_top_k=20)
    assert r['anchor_index'].tolist() == [0, 2] and r['classes'].tolist() == [1, 2]
    assert np.allclose(r['scores'], [.8 * .99, .7 * .97])
    # keep_top_k = 1 stops after the first pick
    r = rp.post_eval_image([pred], [obj], [box], (320, 320), keep_top_k=1)
    assert r['anchor_index'].tolist() == [0]
    # a larger original image raises min_size (0.03 * sqrt(4)) and removes the 0.3-wide box
    r = rp.post_eval_image([pred], [obj], [box], (640, 640), nms_threshold=0.4)
    assert r['anchor_index'].tolist() == [0] or r['anchor_index'].tolist() == [0, 2]
    assert abs(rp.filter_min_size((640, 640)) - 0.06) < 1e-7 and rp.filter_min_size((1, 1)) == np.float32(0.0001)


def test_by_class_variant_hand_case():
    """tf_bboxes_nms_by_class_v1 (ron_eval.py:282-366): a kept box suppresses boxes of ITS label only; the result is cut to the first
    keep_top_k kept rows in score order."""
    pred = np.array([[.1, .8, .1],      # label 1, kept
                     [.1, .7, .2],      # label 1, overlaps anchor 0 -> suppressed in both variants
                     [.1, .2, .7],      # label 2, the SAME box as anchor 0: suppressed by the class-agnostic NMS, kept by class
                     [.2, .1, .7]],     # label 2, far away
                    np.float32).reshape(1, 1, 4, 3)
    obj = np.array([.99, .98, .97, .96], np.float32).reshape(1, 1, 4, 1)
    box = np.array([[.1, .1, .5, .5], [.12, .1, .5, .5], [.1, .1, .5, .5], [.6, .6, .9, .9]], np.float32).reshape(1, 1, 4, 4)
    agnostic = rp.post_eval_image([pred], [obj], [box], (320, 320))
    assert agnostic['anchor_index'].tolist() == [0, 3]
    by_class = rp.post_eval_image([pred], [obj], [box], (320, 320), nms_by_class=True)
    assert by_class['anchor_index'].tolist() == [0, 2, 3] and by_class['classes'].tolist() == [1, 2, 2]
    assert rp.post_eval_image([pred], [obj], [box], (320, 320), nms_by_class=True, keep_top_k=2)['anchor_index'].tolist() == [0, 2]


def test_by_class_scores_variant_hand_case():
    """tf_bboxes_nms_by_class (ron_eval.py:212-280): one greedy NMS per score COLUMN (background included) over every row whose score in
    that column passes select_threshold; a row comes back with the largest score a column kept it with, and that column as its label."""
    pred = np.array([[.10, .50, .40],     # r0, box A: kept by column 1
                     [.10, .45, .45],     # r1, box A': column 1 drops it (overlaps r0), column 2 keeps it -> comes back as label 2
                     [.05, .15, .80],     # r2, box B: column 2
                     [.40, .45, .15],     # r3, box C: column 1 drops it (r4 overlaps), the BACKGROUND column keeps it -> label 0
                     [.30, .55, .15]],    # r4, box C'
                    np.float32).reshape(1, 1, 5, 3)
    obj = np.array([.99, .98, .97, .96, .995], np.float32).reshape(1, 1, 5, 1)
    box = np.array([[.1, .1, .5, .5], [.12, .1, .5, .5], [.6, .6, .9, .9], [.1, .6, .4, .9], [.11, .6, .4, .9]], np.float32).reshape(1, 1, 5, 4)
    r = rp.post_eval_image([pred], [obj], [box], (320, 320), objectness_thres=0.95, select_threshold=0.3, nms_threshold=0.4,
                           keep_top_k=20, nms_by_class='scores')
    assert r['anchor_index'].tolist() == [0, 1, 2, 3, 4]                      # input (anchor) order, not score order
    assert r['classes'].tolist() == [1, 2, 2, 0, 1]
    f = np.float32
    assert np.array_equal(r['scores'], np.array([f(.99) * f(.5), f(.98) * f(.45), f(.97) * f(.8), f(.96) * f(.4), f(.995) * f(.55)], f))
    # one pick per column: column 0 keeps r3, column 1 its best row r4, column 2 r2
    r = rp.post_eval_image([pred], [obj], [box], (320, 320), select_threshold=0.3, keep_top_k=1, nms_by_class='scores')
    assert r['anchor_index'].tolist() == [2, 3, 4] and r['classes'].tolist() == [2, 0, 1]
    # at the reference's own threshold (0.6) only r2's column-2 score passes
    r = rp.post_eval_image([pred], [obj], [box], (320, 320), nms_by_class='scores')
    assert r['anchor_index'].tolist() == [2] and r['classes'].tolist() == [2]
    # a row kept by two columns takes the larger score (:268-270)
    pred2 = np.array([[.05, .50, .45]], np.float32).reshape(1, 1, 1, 3)
    r = rp.post_eval_image([pred2], [obj[:, :, :1]], [box[:, :, :1]], (320, 320), select_threshold=0.3, nms_by_class='scores')
    assert r['classes'].tolist() == [1] and np.array_equal(r['scores'], [f(.99) * f(.5)])


def test_bboxes_filter_min_hand_case():
    """RONNet.bboxes_filter_min (nets/ron_vgg_320.py:217-233): order-preserving mask, zero padding up to top_k, never truncating."""
    from oracle import tfe_post
    s = np.array([[.9, .8, .7, .6]], np.float32)
    b = np.array([[[.1, .1, .5, .5], [.1, .1, .12, .5], [.2, .2, .6, .22], [.3, .3, .4, .4]]], np.float32)   # 2nd too flat, 3rd too narrow
    os_, ob = tfe_post.bboxes_filter_min(s, b, 3)
    assert os_.shape == (1, 3) and ob.shape == (1, 3, 4)
    assert os_[0].tolist() == [np.float32(.9), np.float32(.6), 0.0] and np.array_equal(ob[0, 1], b[0, 3]) and not ob[0, 2].any()
    os_, ob = tfe_post.bboxes_filter_min(s, b, 1)                              # more rows pass than top_k: all of them stay
    assert os_.shape == (1, 2)
    os_, ob = tfe_post.bboxes_filter_min(s, b, 2, minsize=0.5)                 # nothing passes: top_k zero rows
    assert os_.shape == (1, 2) and not os_.any() and not ob.any()
