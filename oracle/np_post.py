"""Oracle: numpy post-processing (test infrastructure, see ``oracle/__init__.py``).

Restates the numpy path of the reference, ``nets/np_methods.py``:

  decode   ``ssd_bboxes_decode``        np_methods.py:23-53
  select   ``ssd_bboxes_select[_layer]`` np_methods.py:56-131
  sort     ``bboxes_sort``              np_methods.py:137-150
  clip     ``bboxes_clip``              np_methods.py:153-164
  resize   ``bboxes_resize``            np_methods.py:167-183
  IoU      ``bboxes_jaccard``           np_methods.py:186-205
  NMS      ``bboxes_nms``               np_methods.py:229-242

plus the two TF element-wise steps the RON evaluation puts in front of it:
softmax (``nets/ron_vgg_320.py:572-576``) and the objectness gate
(``eval_ron_network.py:227-229``).  Call order of the pipeline follows
``notebooks/ssd_notebook.ipynb`` cell 8: select -> clip -> sort(400) -> nms -> resize.

Differences from the reference that are deliberate and documented:

* every function works per image on a batch (the reference decode broadcasts only
  for batch 1); a batch is a list of independent single-image problems;
* select/sort/nms carry an ``anchor index`` side channel (flat index into the
  21 250-anchor list, layers concatenated coarse->fine) because parity on "box
  indices" is graded on it; the reference drops indices;
* ``bboxes_sort`` uses a *stable* descending sort, i.e. order (score desc,
  candidate position asc).  ``np.argsort(-scores)`` in the reference is an
  unstable introsort, so its order inside a group of exactly equal scores is
  unspecified; for inputs without score ties the two agree exactly.

All arithmetic is float32 with one rounding per operation (no fused
multiply-add), exactly as numpy evaluates the reference expressions.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# element-wise head post-ops (TF side of the reference)
# --------------------------------------------------------------------------- #
def softmax_last(x):
    """slim.softmax over the last axis (nets/ron_vgg_320.py:572,574), float32."""
    x = np.asarray(x, dtype=F32)
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m, dtype=F32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


def objectness_from_logits(objness_logits):
    """softmax over the (neg, pos) pair, keep the positive channel.

    nets/ron_vgg_320.py:574-576 -> shape [..., A, 1].
    """
    return softmax_last(objness_logits)[..., 1:2]


def objectness_gate(predictions, objness_pred, objectness_thres=0.03):
    """eval_ron_network.py:227-229: pred * float(objness > thr), per layer."""
    out = []
    for p, o in zip(predictions, objness_pred):
        out.append((np.asarray(o, F32) > F32(objectness_thres)).astype(F32) * np.asarray(p, F32))
    return out


# --------------------------------------------------------------------------- #
# decode                                                        np_methods.py:23-53
# --------------------------------------------------------------------------- #
def bboxes_decode_layer(feat_localizations, anchor_bboxes,
                        prior_scaling=(0.1, 0.1, 0.2, 0.2)):
    """[B,H,W,A,4] (cx,cy,w,h offsets) -> [B,H,W,A,4] (ymin,xmin,ymax,xmax)."""
    loc = np.asarray(feat_localizations, dtype=F32)
    shape = loc.shape
    n_anchor = shape[-2]
    yref, xref, href, wref = anchor_bboxes
    cells = int(np.prod(np.shape(yref)))
    loc = loc.reshape(-1, cells, n_anchor, 4)              # [B, H*W, A, 4]
    xref = np.reshape(xref, (1, cells, 1)).astype(F32)
    yref = np.reshape(yref, (1, cells, 1)).astype(F32)
    href = np.asarray(href, F32).reshape(1, 1, n_anchor)
    wref = np.asarray(wref, F32).reshape(1, 1, n_anchor)
    ps = [F32(v) for v in prior_scaling]
    cx = loc[..., 0] * wref * ps[0] + xref
    cy = loc[..., 1] * href * ps[1] + yref
    w = wref * np.exp(loc[..., 2] * ps[2], dtype=F32)
    h = href * np.exp(loc[..., 3] * ps[3], dtype=F32)
    out = np.empty_like(loc)
    half = F32(2.0)
    out[..., 0] = cy - h / half
    out[..., 1] = cx - w / half
    out[..., 2] = cy + h / half
    out[..., 3] = cx + w / half
    return out.reshape(shape)


def bboxes_decode(localisations, anchors, prior_scaling=(0.1, 0.1, 0.2, 0.2)):
    """List version (nets/ssd_common.py:477-498 has the same arithmetic)."""
    return [bboxes_decode_layer(l, a, prior_scaling) for l, a in zip(localisations, anchors)]


# --------------------------------------------------------------------------- #
# select                                                      np_methods.py:56-131
# --------------------------------------------------------------------------- #
def bboxes_select_image(predictions_img, bboxes_img, select_threshold):
    """One image.  ``predictions_img``: list of [H,W,A,C]; ``bboxes_img``: list of
    decoded [H,W,A,4].  Returns (classes int64, scores f32, bboxes f32 [K,4],
    anchor_index int64) in the reference's order: layers in list order, then
    anchor-major / class-minor (the row-major order of ``np.where``).
    """
    argmax = select_threshold is None or select_threshold == 0       # np_methods.py:82-89: "score > no-label" criterion
    thr = None if argmax else F32(select_threshold)
    cls_l, sc_l, bb_l, ai_l = [], [], [], []
    base = 0
    for pred, box in zip(predictions_img, bboxes_img):
        n_cls = pred.shape[-1]
        flat = np.asarray(pred, F32).reshape(-1, n_cls)
        boxes = np.asarray(box, F32).reshape(-1, 4)
        if argmax:
            # ONE candidate per anchor: the arg-max class over ALL classes (first maximum, like np.argmax), kept when it
            # is not the background class; anchors the objectness gate zeroed have class 0 and drop out
            c = np.argmax(flat, axis=1)
            anchor = np.nonzero(c > 0)[0]
            cls_l.append(c[anchor].astype(np.int64))
            sc_l.append(flat[anchor, c[anchor]])
        else:
            fg = flat[:, 1:]
            anchor, c = np.nonzero(fg > thr)
            cls_l.append(c.astype(np.int64) + 1)
            sc_l.append(fg[anchor, c])
        bb_l.append(boxes[anchor])
        ai_l.append(anchor.astype(np.int64) + base)
        base += flat.shape[0]
    return (np.concatenate(cls_l), np.concatenate(sc_l),
            np.concatenate(bb_l, axis=0), np.concatenate(ai_l))


# --------------------------------------------------------------------------- #
# clip / sort / resize                                      np_methods.py:137-183
# --------------------------------------------------------------------------- #
def bboxes_clip(bbox_ref, bboxes):
    """max/min against the reference box only; no ymin<=ymax repair (np_methods.py:153-164)."""
    ref = np.asarray(bbox_ref, F32)
    out = np.array(bboxes, dtype=F32, copy=True).reshape(-1, 4)
    out[:, 0] = np.maximum(out[:, 0], ref[0])
    out[:, 1] = np.maximum(out[:, 1], ref[1])
    out[:, 2] = np.minimum(out[:, 2], ref[2])
    out[:, 3] = np.minimum(out[:, 3], ref[3])
    return out


def sort_order(scores, top_k):
    """Indices of the top_k scores: score descending, position ascending on ties."""
    return np.argsort(-np.asarray(scores, F32), kind='stable')[:top_k]


def bboxes_sort(classes, scores, bboxes, top_k=400, extra=None):
    order = sort_order(scores, top_k)
    out = (classes[order], scores[order], bboxes[order])
    if extra is not None:
        out = out + (extra[order],)
    return out


def bboxes_resize(bbox_ref, bboxes):
    """np_methods.py:167-183 (identity for bbox_ref = [0,0,1,1] up to -0.0/0.0)."""
    ref = np.asarray(bbox_ref, F32)
    out = np.array(bboxes, dtype=F32, copy=True).reshape(-1, 4)
    out[:, 0] -= ref[0]
    out[:, 1] -= ref[1]
    out[:, 2] -= ref[0]
    out[:, 3] -= ref[1]
    sy = ref[2] - ref[0]
    sx = ref[3] - ref[1]
    out[:, 0] /= sy
    out[:, 1] /= sx
    out[:, 2] /= sy
    out[:, 3] /= sx
    return out


# --------------------------------------------------------------------------- #
# IoU + greedy class-aware NMS                              np_methods.py:186-242
# --------------------------------------------------------------------------- #
def bboxes_jaccard(box, others):
    """IoU of one box [4] with many [K,4]; plain float32 division (0/0 -> NaN)."""
    box = np.asarray(box, F32)
    others = np.asarray(others, F32).reshape(-1, 4)
    zero = F32(0.0)
    ih = np.maximum(np.minimum(box[2], others[:, 2]) - np.maximum(box[0], others[:, 0]), zero)
    iw = np.maximum(np.minimum(box[3], others[:, 3]) - np.maximum(box[1], others[:, 1]), zero)
    inter = ih * iw
    vol1 = (box[2] - box[0]) * (box[3] - box[1])
    vol2 = (others[:, 2] - others[:, 0]) * (others[:, 3] - others[:, 1])
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / (vol1 + vol2 - inter)


def nms_keep_mask(classes, scores, bboxes, nms_threshold=0.45):
    """Boolean keep mask of the greedy scan (np_methods.py:229-242).

    A later box j is dropped by a kept box i when NOT(IoU < thr) and the classes
    match; a NaN IoU therefore suppresses, as in the reference's
    ``logical_or(overlap < thr, classes != class_i)``.
    """
    n = int(np.shape(scores)[0])
    keep = np.ones((n,), dtype=bool)
    thr = F32(nms_threshold)
    for i in range(n - 1):
        if not keep[i]:
            continue
        iou = bboxes_jaccard(bboxes[i], bboxes[i + 1:])
        survives = (iou < thr) | (classes[i + 1:] != classes[i])
        keep[i + 1:] &= survives
    return keep


def bboxes_nms(classes, scores, bboxes, nms_threshold=0.45, extra=None):
    keep = nms_keep_mask(classes, scores, bboxes, nms_threshold)
    out = (classes[keep], scores[keep], bboxes[keep])
    if extra is not None:
        out = out + (extra[keep],)
    return out


# --------------------------------------------------------------------------- #
# whole pipeline, one batch
# --------------------------------------------------------------------------- #
def detect_from_predictions(predictions, localisations, anchors, objness_pred=None,
                            objectness_thres=0.03, select_threshold=0.01,
                            top_k=400, nms_threshold=0.45,
                            bbox_img=(0., 0., 1., 1.),
                            prior_scaling=(0.1, 0.1, 0.2, 0.2), decode=True):
    """np_methods pipeline for a batch of images.

    predictions[i]   [B,H,W,A,C] softmax scores, localisations[i] [B,H,W,A,4] raw
    offsets (or decoded boxes if ``decode=False``), objness_pred[i] [B,H,W,A,1] or
    None (SSD: no gate).  Returns a list (one entry per image) of dicts with
    ``classes`` int64, ``scores`` f32, ``bboxes`` f32 [K,4], ``anchor_index`` int64,
    ``n_candidates`` (after select) and ``n_sorted`` (after top-k).
    """
    if objness_pred is not None:
        predictions = objectness_gate(predictions, objness_pred, objectness_thres)
    boxes = bboxes_decode(localisations, anchors, prior_scaling) if decode else localisations
    batch = predictions[0].shape[0]
    results = []
    for b in range(batch):
        cls, sc, bb, ai = bboxes_select_image([p[b] for p in predictions],
                                              [x[b] for x in boxes], select_threshold)
        n_cand = cls.shape[0]
        bb = bboxes_clip(bbox_img, bb)
        cls, sc, bb, ai = bboxes_sort(cls, sc, bb, top_k=top_k, extra=ai)
        n_sorted = cls.shape[0]
        cls, sc, bb, ai = bboxes_nms(cls, sc, bb, nms_threshold, extra=ai)
        bb = bboxes_resize(bbox_img, bb)
        results.append(dict(classes=cls, scores=sc, bboxes=bb, anchor_index=ai,
                            n_candidates=n_cand, n_sorted=n_sorted))
    return results


def detect_from_logits(cls_logits, objness_logits, localisations, anchors, **kw):
    """Same, starting from the raw head tensors the conv stack emits."""
    predictions = [softmax_last(l) for l in cls_logits]
    objness = [objectness_from_logits(o) for o in objness_logits]
    return detect_from_predictions(predictions, localisations, anchors, objness_pred=objness, **kw)
