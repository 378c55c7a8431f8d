"""Oracle: evaluation bookkeeping (test infrastructure, see ``oracle/__init__.py``).

Restates the TF graph code the reference's eval driver runs after the detections:

  matching   ``tfe.bboxes_matching[_batch]``      tf_extended/bboxes.py:316-450   (jaccard :527-555, safe_divide math.py:25-38)
  streaming  ``tfe.streaming_tp_fp_arrays``       tf_extended/metrics.py:133-206
  P / R      ``tfe.precision_recall``             tf_extended/metrics.py:100-130
  AP         ``average_precision_voc07 / voc12``  tf_extended/metrics.py:212-258

TensorFlow code: **parity unpinned** for matching / streaming; the two AP integrals ARE pinned against the reference's own
numpy ``voc_ap`` (datasets/voc_eval.py:130-162, same definitions) through tests/golden/g6_voc_ap.npz.
"""
import numpy as np

F32 = np.float32


def jaccard(box, gboxes):
    """tf_extended/bboxes.py:527-555: float32, 0 where the union is not > 0."""
    box = np.asarray(box, F32)
    g = np.asarray(gboxes, F32).reshape(-1, 4)
    zero = F32(0)
    h = np.maximum(np.minimum(g[:, 2], box[2]) - np.maximum(g[:, 0], box[0]), zero)
    w = np.maximum(np.minimum(g[:, 3], box[3]) - np.maximum(g[:, 1], box[1]), zero)
    inter = h * w
    union = -inter + (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1]) + (box[2] - box[0]) * (box[3] - box[1])
    out = np.zeros_like(inter)
    ok = union > 0
    out[ok] = inter[ok] / union[ok]
    return out


def bboxes_matching(label, scores, bboxes, glabels, gbboxes, gdifficults, matching_threshold=0.5):
    """One image, one class: (n_gbboxes, tp[K] bool, fp[K] bool).  Detections must be sorted by score."""
    glabels = np.asarray(glabels)
    gdiff = np.asarray(gdifficults).astype(bool)
    n_gb = int(np.count_nonzero((glabels == label) & ~gdiff))
    gmatch = np.zeros(glabels.shape, bool)
    k = scores.shape[0]
    tp = np.zeros((k,), bool)
    fp = np.zeros((k,), bool)
    same = (glabels == label).astype(F32)
    thr = F32(matching_threshold)
    for i in range(k):
        jac = jaccard(bboxes[i], gbboxes) * same
        idx = int(np.argmax(jac))                      # first maximum, like tf.argmax
        match = jac[idx] > thr
        existing = gmatch[idx]
        not_diff = not gdiff[idx]
        tp[i] = not_diff and match and not existing
        fp[i] = not_diff and (existing or not match)
        if not_diff and match:
            gmatch[idx] = True
    return n_gb, tp, fp


def bboxes_matching_batch(labels, d_scores, d_bboxes, glabels, gbboxes, gdifficults, matching_threshold=0.5):
    """Dict form over classes and batch: class -> (n_gbboxes [B], tp [B,K], fp [B,K])."""
    d_n, d_tp, d_fp = {}, {}, {}
    for c in labels:
        b = d_scores[c].shape[0]
        ns, tps, fps = [], [], []
        for i in range(b):
            n, tp, fp = bboxes_matching(c, d_scores[c][i], d_bboxes[c][i], glabels[i], gbboxes[i], gdifficults[i], matching_threshold)
            ns.append(n); tps.append(tp); fps.append(fp)
        d_n[c], d_tp[c], d_fp[c] = np.array(ns, np.int64), np.stack(tps), np.stack(fps)
    return d_n, d_tp, d_fp


def streaming_filter(tp, fp, scores, remove_zero_scores=True):
    """tf_extended/metrics.py:170-181: flatten, keep entries with tp|fp (and score > 1e-4)."""
    tp, fp, scores = np.ravel(tp).astype(bool), np.ravel(fp).astype(bool), np.ravel(scores).astype(F32)
    mask = tp | fp
    if remove_zero_scores:
        mask = mask & (scores > F32(1e-4))
        return tp[mask], fp[mask], scores[mask]
    return tp, fp, scores              # the reference only applies the mask inside the branch


def precision_recall(num_gbboxes, tp, fp, scores):
    """tf_extended/metrics.py:121-130 in float64: sort by score (descending, lower index first on ties), cumulate."""
    order = np.argsort(-np.asarray(scores, F32), kind='stable')
    ctp = np.cumsum(tp[order].astype(np.float64))
    cfp = np.cumsum(fp[order].astype(np.float64))
    recall = ctp / float(num_gbboxes) if num_gbboxes > 0 else np.zeros_like(ctp)
    den = ctp + cfp
    precision = np.where(den > 0, ctp / np.where(den > 0, den, 1.0), 0.0)
    return precision, recall


def average_precision_voc07(precision, recall):
    """11-point metric, tf_extended/metrics.py:238-258."""
    p = np.concatenate([np.asarray(precision, np.float64), [0.0]])
    r = np.concatenate([np.asarray(recall, np.float64), [np.inf]])
    ap = 0.0
    for t in np.arange(0., 1.1, 0.1):
        ap += p[r >= t].max() / 11.0
    return ap


def average_precision_voc12(precision, recall):
    """Area under the monotone envelope, tf_extended/metrics.py:212-235."""
    p = np.concatenate([[0.0], np.asarray(precision, np.float64), [0.0]])
    r = np.concatenate([[0.0], np.asarray(recall, np.float64), [1.0]])
    p = np.maximum.accumulate(p[::-1])[::-1]
    return float(np.sum(p[1:] * (r[1:] - r[:-1])))


def voc_eval_class(image_ids, confidence, boxes, gt_boxes, gt_difficult, ovthresh=0.5, use_07_metric=True):
    """datasets/voc_eval.py:224-295 (one class) on arrays: **pinned** by tests/golden/g7_voc_eval.npz, which the reference's own
    ``DetectorEvalPascal.voc_eval`` produced from files holding these arrays."""
    n = len(image_ids)
    if n == 0:
        return -1., -1., -1.
    conf = np.asarray(confidence, np.float64)
    bbs = np.asarray(boxes, np.float64).reshape(-1, 4)
    npos = sum(int((~np.asarray(d, bool)).sum()) for d in gt_difficult.values())
    used = {k: [False] * len(v) for k, v in gt_difficult.items()}
    order = np.argsort(-conf)
    tp, fp = np.zeros(n), np.zeros(n)
    for d, o in enumerate(order):
        img, bb = int(image_ids[o]), bbs[o]
        g = np.asarray(gt_boxes.get(img, []), np.float64).reshape(-1, 4)
        ovmax, jmax = -np.inf, -1
        if g.size:
            iw = np.maximum(np.minimum(g[:, 2], bb[2]) - np.maximum(g[:, 0], bb[0]), 0.)
            ih = np.maximum(np.minimum(g[:, 3], bb[3]) - np.maximum(g[:, 1], bb[1]), 0.)
            inter = iw * ih
            ov = inter / ((bb[2] - bb[0]) * (bb[3] - bb[1]) + (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1]) - inter)
            ovmax, jmax = ov.max(), int(ov.argmax())
        if ovmax > ovthresh:
            if not gt_difficult[img][jmax]:
                if not used[img][jmax]:
                    tp[d] = 1.
                    used[img][jmax] = True
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, (average_precision_voc07 if use_07_metric else average_precision_voc12)(prec, rec)
