"""Oracle: post-processing of ``ron_eval.py`` (test infrastructure, see ``oracle/__init__.py``).

  flaten_predict   ron_eval.py:111-144   score = objectness * class probability, label = argmax over all classes,
                                         kept when label > 0 and objectness > objectness_thres
  clip             tfe.bboxes_clip       tf_extended/bboxes.py:105-144 (oracle/tfe_post.clip_with_repair)
  filter_boxes     ron_eval.py:369-392   sides > min_size, centre strictly inside (0, 1)
  tf_bboxes_nms    ron_eval.py:146-206   score > select_threshold, sort (tf.nn.top_k: lower index first among equals),
                                         greedy, all classes together, at most keep_top_k kept
  tf_bboxes_nms_by_class_v1   ron_eval.py:282-366   (the variant behind the commented call of :474) per label: greedy among the rows of
                                         that label; the union, cut to the first keep_top_k kept rows in score order
  tf_bboxes_nms_by_class      ron_eval.py:212-280   (the function that commented call names) per score COLUMN, background included:
                                         rows with score[c] > select_threshold sorted by score[c], greedy with at most keep_top_k picks;
                                         a row any column kept is returned with max / argmax of its kept scores, rows in input order
  resize           tfe.bboxes_resize     tf_extended/bboxes.py:147-171

TensorFlow graph code: **parity unpinned** (hand case in tests/test_oracle_ron_eval.py).
"""
import numpy as np

from . import np_post
from . import tfe_post

F32 = np.float32


def filter_min_size(image_hw, net_input_shape=(320., 320.), min_size_ratio=0.03):
    h, w = int(image_hw[0]), int(image_hw[1])
    return max(F32(0.0001), F32(min_size_ratio) * np.sqrt(F32(h * w) / F32(net_input_shape[0] * net_input_shape[1])))


def flaten_predict(predictions, objness_pred, bboxes, objectness_thres):
    """One image: lists of [H,W,A,C] / [H,W,A,1] / [H,W,A,4] -> (scores [M] = max_c, labels [M], bboxes [M,4], anchor_index [M])."""
    C = predictions[0].shape[-1]
    pred = np.concatenate([np.asarray(p, F32).reshape(-1, C) for p in predictions], 0)
    obj = np.concatenate([np.asarray(o, F32).reshape(-1) for o in objness_pred], 0)
    box = np.concatenate([np.asarray(b, F32).reshape(-1, 4) for b in bboxes], 0)
    cls_pred = obj[:, None] * pred
    labels = np.argmax(cls_pred, -1)
    mask = (labels > 0) & (obj > F32(objectness_thres))
    idx = np.flatnonzero(mask)
    return cls_pred[idx].max(-1), labels[idx], box[idx], idx


def flaten_predict_columns(predictions, objness_pred, bboxes, objectness_thres):
    """flaten_predict as main() uses it (ron_eval.py:466): scores [M, C] = objectness * every class probability, with the same mask."""
    C = predictions[0].shape[-1]
    pred = np.concatenate([np.asarray(p, F32).reshape(-1, C) for p in predictions], 0)
    obj = np.concatenate([np.asarray(o, F32).reshape(-1) for o in objness_pred], 0)
    box = np.concatenate([np.asarray(b, F32).reshape(-1, 4) for b in bboxes], 0)
    cls_pred = obj[:, None] * pred
    labels = np.argmax(cls_pred, -1)
    mask = (labels > 0) & (obj > F32(objectness_thres))
    idx = np.flatnonzero(mask)
    return cls_pred[idx], labels[idx], box[idx], idx


def filter_boxes(scores, labels, bboxes, extra, min_size):
    ws = bboxes[:, 3] - bboxes[:, 1]
    hs = bboxes[:, 2] - bboxes[:, 0]
    xc = bboxes[:, 1] + ws / F32(2.)
    yc = bboxes[:, 0] + hs / F32(2.)
    keep = (ws > min_size) & (hs > min_size) & (xc > 0) & (yc > 0) & (xc < 1) & (yc < 1)
    return scores[keep], labels[keep], bboxes[keep], extra[keep]


def tf_bboxes_nms(scores, labels, bboxes, extra, select_threshold, nms_threshold, keep_top_k, mode):
    m = scores > F32(select_threshold)
    scores, labels, bboxes, extra = scores[m], labels[m], bboxes[m], extra[m]
    if scores.shape[0] < 1:
        return scores, labels, bboxes, extra
    order = np.argsort(-scores, kind='stable')
    scores, labels, bboxes, extra = scores[order], labels[order], bboxes[order], extra[order]
    n = scores.shape[0]
    alive = np.ones((n,), bool)
    keep = np.zeros((n,), bool)
    it = 0
    while alive.any() and it < keep_top_k:
        i = int(np.flatnonzero(alive)[0])
        keep[i] = True
        alive[i] = False
        ov = tfe_post.overlap_scores(bboxes[i], bboxes, mode) * alive.astype(F32)
        alive &= ov < F32(nms_threshold)
        it += 1
    return scores[keep], labels[keep], bboxes[keep], extra[keep]


def tf_bboxes_nms_by_class_v1(scores, labels, bboxes, extra, select_threshold, nms_threshold, keep_top_k, mode, num_classes=21):
    """ron_eval.py:282-366, written like the reference: a while loop over the labels 1 .. num_classes - 1, each a greedy loop of at
    most keep_top_k picks over the rows of that label; then the cut at the (keep_top_k + 1)-th kept row (:358-361)."""
    m = scores > F32(select_threshold)
    scores, labels, bboxes, extra = scores[m], labels[m], bboxes[m], extra[m]
    if scores.shape[0] < 1:
        return scores, labels, bboxes, extra
    order = np.argsort(-scores, kind='stable')
    scores, labels, bboxes, extra = scores[order], labels[order], bboxes[order], extra[order]
    n = scores.shape[0]
    total = np.zeros((n,), bool)
    for c in range(1, num_classes):
        alive = labels == c
        it = 0
        while alive.any() and it < keep_top_k:
            i = int(np.flatnonzero(alive)[0])
            total[i] = True
            alive[i] = False
            ov = tfe_post.overlap_scores(bboxes[i], bboxes, mode) * alive.astype(F32)
            alive &= ov < F32(nms_threshold)
            it += 1
    kept = np.flatnonzero(total)
    if kept.shape[0] >= keep_top_k + 1:
        total &= np.arange(n) < kept[keep_top_k]
    return scores[total], labels[total], bboxes[total], extra[total]


def tf_bboxes_nms_by_class(scores, labels, bboxes, extra, select_threshold, nms_threshold, keep_top_k, mode):
    """ron_eval.py:212-280.  scores [M, C]: nms_proc (:215-258) once per column - tf.nn.top_k over all M rows (stable: lower index first
    among equals), alive = score > select_threshold (:226), greedy (:251-257: pick the first live row in sorted order, mark the ORIGINAL
    row kept, :247-248, drop the live rows it overlaps), at most keep_top_k picks (:240); then (:266-274) keep_scores = scores * mask,
    max / argmax over the columns, rows with max > 0, in input order."""
    n, C = scores.shape
    if n < 1:
        return scores.reshape(-1)[:0], labels, bboxes, extra
    total = np.zeros((n, C), bool)
    for c in range(C):
        col = scores[:, c]
        order = np.argsort(-col, kind='stable')
        sb = bboxes[order]
        alive = col[order] > F32(select_threshold)
        it = 0
        while alive.any() and it < keep_top_k:
            i = int(np.flatnonzero(alive)[0])
            total[order[i], c] = True
            alive[i] = False
            ov = tfe_post.overlap_scores(sb[i], sb, mode) * alive.astype(F32)
            alive &= ov < F32(nms_threshold)
            it += 1
    keep_scores = scores * total.astype(F32)
    mx = keep_scores.max(-1)
    new_labels = np.argmax(keep_scores, -1)
    keep = mx > 0
    return mx[keep], new_labels[keep], bboxes[keep], extra[keep]


def post_eval_image(predictions, objness_pred, bboxes, image_hw, objectness_thres=0.95, select_threshold=0.6, nms_threshold=0.4,
                    keep_top_k=20, nms_mode='union', bbox_img=(0., 0., 1., 1.), min_size_ratio=0.03, nms_by_class=False):
    """ron_eval.py:466-477 for one image (decoded boxes in): dict classes / scores / bboxes / anchor_index."""
    if nms_by_class == 'scores':
        s, l, b, idx = flaten_predict_columns(predictions, objness_pred, bboxes, objectness_thres)
    else:
        s, l, b, idx = flaten_predict(predictions, objness_pred, bboxes, objectness_thres)
    b = tfe_post.clip_with_repair(bbox_img, b)
    s, l, b, idx = filter_boxes(s, l, b, idx, filter_min_size(image_hw, min_size_ratio=min_size_ratio))
    if nms_by_class == 'scores':
        s, l, b, idx = tf_bboxes_nms_by_class(s, l, b, idx, select_threshold, nms_threshold, keep_top_k, nms_mode)
    elif nms_by_class:
        s, l, b, idx = tf_bboxes_nms_by_class_v1(s, l, b, idx, select_threshold, nms_threshold, keep_top_k, nms_mode, predictions[0].shape[-1])
    else:
        s, l, b, idx = tf_bboxes_nms(s, l, b, idx, select_threshold, nms_threshold, keep_top_k, nms_mode)
    b = np_post.bboxes_resize(bbox_img, b)
    return dict(classes=l.astype(np.int64), scores=s, bboxes=b, anchor_index=idx.astype(np.int64))
