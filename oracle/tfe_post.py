"""Oracle: the TF-evaluation post-processing variant (test infrastructure, see ``oracle/__init__.py``).

Restates, in numpy, what ``eval_ron_network.py:226-236`` runs on the CPU device through
``RONNet.detected_bboxes`` (``nets/ron_vgg_320.py:234-256``):

  select      ``ssd_common.tf_ssd_bboxes_select[_layer]``   nets/ssd_common.py:504-589
  clip        ``tfe.bboxes_clip``                           tf_extended/bboxes.py:105-144  (with ymin<=ymax repair)
  filter_min  ``RONNet.bboxes_filter_min``                  nets/ron_vgg_320.py:196-233
  sort        ``tfe.bboxes_sort`` (tf.nn.top_k)             tf_extended/bboxes.py:60-101
  nms         ``tfe.bboxes_nms`` / ``bboxes_nms_batch``     tf_extended/bboxes.py:173-234, :262-302
  zero pad    ``tfe.pad_axis``                              tf_extended/tensors.py:59-86

These are TensorFlow-1 graph functions; TensorFlow cannot be installed here and the reference holds no
test or golden vector for them: **parity unpinned** (hand-derived cases in tests/test_oracle_tfe.py).

Semantics restated:
* per class c = 1..C-1 the score/box lists are dense over all anchors; entries with score <= thr are zeroed
  (score AND box), so they have zero width/height and are removed by ``bboxes_filter_min`` (w > minsize and
  h > minsize, order preserving) whenever minsize >= 0;  for SSD (no filter) they stay as zero rows;
* ``tf.nn.top_k`` sorts by score descending, lower index first among equal scores; lists shorter than
  top_k are zero padded first;
* NMS: greedy in sorted order, at most keep_top_k boxes are kept; a later box j is dropped by a kept box i
  when NOT(overlap(i, j) < thr) with overlap = inter / min(area_j, area_i) (mode 'min', the default) or
  inter / (area_j - inter + area_i) (mode 'union'), and 0 where the denominator is not > 0
  (``safe_divide``, tf_extended/bboxes.py:192-193).  Zero (padding) rows are never suppressed and never
  suppress; kept zero rows are indistinguishable from the zero padding of the output.
"""
import numpy as np

F32 = np.float32


def clip_with_repair(bbox_ref, bboxes):
    """tf_extended/bboxes.py:129-143."""
    ref = np.asarray(bbox_ref, F32)
    b = np.asarray(bboxes, F32).reshape(-1, 4)
    ymin = np.maximum(b[:, 0], ref[0])
    xmin = np.maximum(b[:, 1], ref[1])
    ymax = np.minimum(b[:, 2], ref[2])
    xmax = np.minimum(b[:, 3], ref[3])
    ymin = np.minimum(ymin, ymax)
    xmin = np.minimum(xmin, xmax)
    return np.stack([ymin, xmin, ymax, xmax], axis=1)


def overlap_scores(box, boxes, mode):
    """tf_extended/bboxes.py:195-211 (without the mask factor)."""
    zero = F32(0)
    ih = np.maximum(np.minimum(boxes[:, 2], box[2]) - np.maximum(boxes[:, 0], box[0]), zero)
    iw = np.maximum(np.minimum(boxes[:, 3], box[3]) - np.maximum(boxes[:, 1], box[1]), zero)
    inner = ih * iw
    this_vol = (box[2] - box[0]) * (box[3] - box[1])
    vol = (boxes[:, 3] - boxes[:, 1]) * (boxes[:, 2] - boxes[:, 0])
    if mode == 'union':
        den = vol - inner + this_vol
    elif mode == 'min':
        den = np.minimum(vol, this_vol)
    else:
        raise ValueError('unknown mode to use for nms.')
    out = np.zeros_like(inner)
    ok = den > 0
    out[ok] = inner[ok] / den[ok]
    return out


def bboxes_filter_min(scores, bboxes, top_k, minsize=0.03):
    """RONNet.bboxes_filter_min on tensors (nets/ron_vgg_320.py:217-233): scores [1, N], bboxes [1, N, 4] -> squeeze axis 0, keep the
    rows with w > minsize and h > minsize (tf.boolean_mask: order preserving), pad_axis(…, 0, top_k) = zeros up to top_k rows, never
    shorter than what passed (tf_extended/tensors.py:59-86), expand_dims back."""
    s, b = np.asarray(scores, F32)[0], np.asarray(bboxes, F32)[0]
    h = b[:, 2] - b[:, 0]
    w = b[:, 3] - b[:, 1]
    m = (w > F32(minsize)) & (h > F32(minsize))
    s, b = s[m], b[m]
    pad = max(int(top_k) - s.shape[0], 0)
    s = np.concatenate([s, np.zeros((pad,), F32)])
    b = np.concatenate([b, np.zeros((pad, 4), F32)])
    return s[None], b[None]


def nms_one_class(scores, bboxes, nms_threshold, keep_top_k, mode='min'):
    """Sorted lists in, (scores [keep_top_k], bboxes [keep_top_k, 4]) zero padded out."""
    n = scores.shape[0]
    alive = np.ones((n,), dtype=bool)
    keep = np.zeros((n,), dtype=bool)
    thr = F32(nms_threshold)
    it = 0
    while alive.any() and it < keep_top_k:
        i = int(np.flatnonzero(alive)[0])
        keep[i] = True
        alive[i] = False
        ov = overlap_scores(bboxes[i], bboxes, mode) * alive.astype(F32)
        alive &= ov < thr
        it += 1
    out_s = np.zeros((max(keep_top_k, int(keep.sum())),), F32)
    out_b = np.zeros((out_s.shape[0], 4), F32)
    k = int(keep.sum())
    out_s[:k] = scores[keep]
    out_b[:k] = bboxes[keep]
    return out_s[:keep_top_k], out_b[:keep_top_k]


def detected_bboxes(predictions, localisations, num_classes=21, select_threshold=None, nms_threshold=0.5,
                    clipping_bbox=None, top_k=400, keep_top_k=200, nms_mode='min', min_size=0.03):
    """predictions[i] [B,H,W,A,C] (already gated), localisations[i] [B,H,W,A,4] DECODED boxes.

    Returns (dict_scores, dict_bboxes): class -> [B, keep_top_k] / [B, keep_top_k, 4] like the reference.
    """
    thr = F32(0.0 if select_threshold is None else select_threshold)
    batch = predictions[0].shape[0]
    pred = np.concatenate([np.asarray(p, F32).reshape(batch, -1, num_classes) for p in predictions], axis=1)
    loc = np.concatenate([np.asarray(l, F32).reshape(batch, -1, 4) for l in localisations], axis=1)
    d_scores, d_bboxes = {}, {}
    for c in range(1, num_classes):
        out_s = np.zeros((batch, keep_top_k), F32)
        out_b = np.zeros((batch, keep_top_k, 4), F32)
        for b in range(batch):
            sc = pred[b, :, c]
            fmask = (sc > thr).astype(F32)
            sc = sc * fmask
            bx = loc[b] * fmask[:, None]
            if clipping_bbox is not None:
                bx = clip_with_repair(clipping_bbox, bx)
            if min_size is not None and min_size >= 0:
                h = bx[:, 2] - bx[:, 0]
                w = bx[:, 3] - bx[:, 1]
                m = (w > F32(min_size)) & (h > F32(min_size))
                sc, bx = sc[m], bx[m]
            if sc.shape[0] < top_k:           # pad_axis(..., top_k)
                pad = top_k - sc.shape[0]
                sc = np.concatenate([sc, np.zeros((pad,), F32)])
                bx = np.concatenate([bx, np.zeros((pad, 4), F32)], axis=0)
            order = np.argsort(-sc, kind='stable')[:top_k]          # tf.nn.top_k: lower index first on ties
            sc, bx = sc[order], bx[order]
            s, bb = nms_one_class(sc, bx, nms_threshold, keep_top_k, nms_mode)
            out_s[b], out_b[b] = s, bb
        d_scores[c], d_bboxes[c] = out_s, out_b
    return d_scores, d_bboxes
