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
