"""Evaluation bookkeeping behind the detections: what eval_ron_network.py:237-335 builds from
``tfe.bboxes_matching_batch``, ``tfe.streaming_tp_fp_arrays``, ``tfe.precision_recall`` and
``tfe.average_precision_voc07 / voc12`` (tf_extended/bboxes.py:316-450, tf_extended/metrics.py:100-258).

The matching runs on the GPU (``ron_bboxes_matching``, one wave per (image, class) list); the streaming accumulators
and the P/R/AP integrals are a few thousand scalars per class and stay on the host in float64, as the reference
computes them."""
import numpy as np
import torch

from ._lib import check, current_stream, lib, ptr


def bboxes_matching(scores, bboxes, glabels, gbboxes, gdifficults, matching_threshold=0.5):
    """Dense form: scores [N, L, K], bboxes [N, L, K, 4] (list l = class l + 1, sorted by score, as ron_post_tfe
    writes them); glabels [N, G], gbboxes [N, G, 4], gdifficults [N, G].  Returns GPU tensors
    (n_gbboxes int32 [N, L], tp bool [N, L, K], fp bool [N, L, K])."""
    assert scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 3
    n, nl, k = scores.shape
    dev = scores.device
    scores, bboxes = scores.contiguous(), bboxes.contiguous()
    gl = torch.as_tensor(glabels).to(device=dev, dtype=torch.int32).contiguous()
    gb = torch.as_tensor(gbboxes).to(device=dev, dtype=torch.float32).contiguous()
    gd = (torch.as_tensor(gdifficults).to(device=dev) != 0).to(torch.uint8).contiguous()
    g = gl.shape[1]
    assert tuple(bboxes.shape) == (n, nl, k, 4) and tuple(gb.shape) == (n, g, 4) and tuple(gd.shape) == (n, g)
    n_gb = torch.empty((n, nl), dtype=torch.int32, device=dev)
    tp = torch.empty((n, nl, k), dtype=torch.uint8, device=dev)
    fp = torch.empty((n, nl, k), dtype=torch.uint8, device=dev)
    check(lib().ron_bboxes_matching(ptr(scores), ptr(bboxes), n, nl, k, ptr(gl), ptr(gb), ptr(gd), g,
                                    float(matching_threshold), ptr(n_gb), ptr(tp), ptr(fp), current_stream()))
    return n_gb, tp.bool(), fp.bool()


def bboxes_matching_batch(labels, scores, bboxes, glabels, gbboxes, gdifficults, matching_threshold=0.5):
    """Reference signature (tf_extended/bboxes.py:397-450) for the dict case: ``scores`` / ``bboxes`` are dicts
    class -> [N, K] / [N, K, 4]; returns dicts class -> n_gbboxes [N], tp [N, K], fp [N, K]."""
    labels = list(labels)
    s = torch.stack([scores[c] for c in labels], dim=1)
    b = torch.stack([bboxes[c] for c in labels], dim=1)
    if labels != list(range(1, len(labels) + 1)):
        # the kernel derives the label from the list index; remap arbitrary label sets onto 1..L
        gl = torch.as_tensor(glabels).to(s.device)
        remap = torch.zeros_like(gl)
        for i, c in enumerate(labels):
            remap = torch.where(gl == c, torch.full_like(gl, i + 1), remap)
        glabels = remap
    n_gb, tp, fp = bboxes_matching(s, b, glabels, gbboxes, gdifficults, matching_threshold)
    return ({c: n_gb[:, i].to(torch.int64) for i, c in enumerate(labels)},
            {c: tp[:, i] for i, c in enumerate(labels)},
            {c: fp[:, i] for i, c in enumerate(labels)})


class StreamingTpFp(object):
    """tfe.streaming_tp_fp_arrays (tf_extended/metrics.py:133-206) for every class at once: accumulates
    (num_gbboxes, num_detections, tp, fp, scores) over batches."""

    def __init__(self, labels, remove_zero_scores=True):
        self.labels = list(labels)
        self.remove_zero_scores = remove_zero_scores
        self.num_gbboxes = {c: 0 for c in self.labels}
        self.num_detections = {c: 0 for c in self.labels}
        self._tp = {c: [] for c in self.labels}
        self._fp = {c: [] for c in self.labels}
        self._scores = {c: [] for c in self.labels}

    def update(self, n_gbboxes, tp, fp, scores):
        """Dense batch results [N, L], [N, L, K] x 3 (GPU or host); list l belongs to ``labels[l]``."""
        n_gbboxes, tp, fp, scores = (t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
                                     for t in (n_gbboxes, tp, fp, scores))
        for i, c in enumerate(self.labels):
            t, f, s = tp[:, i].reshape(-1).astype(bool), fp[:, i].reshape(-1).astype(bool), scores[:, i].reshape(-1).astype(np.float32)
            if self.remove_zero_scores:                               # metrics.py:170-181
                mask = (t | f) & (s > np.float32(1e-4))
                t, f, s = t[mask], f[mask], s[mask]
            self.num_gbboxes[c] += int(n_gbboxes[:, i].sum())
            self.num_detections[c] += int(t.shape[0])
            self._tp[c].append(t)
            self._fp[c].append(f)
            self._scores[c].append(s)

    def arrays(self, c):
        """(num_gbboxes, num_detections, tp, fp, scores) of class c, the tuple the reference streams."""
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros((0,), dt)
        return (self.num_gbboxes[c], self.num_detections[c], cat(self._tp[c], bool), cat(self._fp[c], bool),
                cat(self._scores[c], np.float32))


def precision_recall(num_gbboxes, num_detections, tp, fp, scores, dtype=np.float64):
    """tfe.precision_recall (metrics.py:100-130): sort by score (all detections), cumulate in float64."""
    scores = np.asarray(scores, np.float32)
    order = np.argsort(-scores, kind='stable')
    ctp = np.cumsum(np.asarray(tp)[order].astype(dtype))
    cfp = np.cumsum(np.asarray(fp)[order].astype(dtype))
    recall = ctp / dtype(num_gbboxes) if num_gbboxes > 0 else np.zeros_like(ctp)      # safe_divide
    den = ctp + cfp
    precision = np.where(den > 0, ctp / np.where(den > 0, den, 1), 0)
    return precision, recall


def average_precision_voc12(precision, recall):
    """Area under the monotone precision envelope (metrics.py:212-235)."""
    p = np.concatenate([[0.], np.asarray(precision, np.float64), [0.]])
    r = np.concatenate([[0.], np.asarray(recall, np.float64), [1.]])
    p = np.maximum.accumulate(p[::-1])[::-1]
    return float(np.sum(p[1:] * (r[1:] - r[:-1])))


def average_precision_voc07(precision, recall):
    """11-point interpolated AP (metrics.py:238-258)."""
    p = np.concatenate([np.asarray(precision, np.float64), [0.]])
    r = np.concatenate([np.asarray(recall, np.float64), [np.inf]])
    ap = 0.
    for t in np.arange(0., 1.1, 0.1):
        ap += float(np.max(p[r >= t])) / 11.
    return ap


def voc_eval_class(image_ids, confidence, boxes, gt_boxes, gt_difficult, ovthresh=0.5, use_07_metric=True):
    """The PASCAL evaluation of ONE class as ``ron_eval.py`` runs it (datasets/voc_eval.py:164-295), on arrays instead of files:
    image_ids [D] int, confidence [D], boxes [D, 4] (x1, y1, x2, y2, pixels) of the detections; gt_boxes / gt_difficult: dicts
    image id -> [G, 4] / [G] of this class's ground truth (as parse_rec returns it: XML values minus one).
    Returns (recall [D], precision [D], ap), or (-1., -1., -1.) when there is no detection, like the reference."""
    image_ids = np.asarray(image_ids).reshape(-1)
    if image_ids.shape[0] == 0:
        return -1., -1., -1.
    confidence = np.asarray(confidence, np.float64).reshape(-1)
    bb_all = np.asarray(boxes, np.float64).reshape(-1, 4)
    npos = int(sum(int(np.sum(~np.asarray(d).astype(bool))) for d in gt_difficult.values()))
    det = {k: np.zeros(len(np.asarray(v).reshape(-1)), bool) for k, v in gt_difficult.items()}
    order = np.argsort(-confidence)
    nd = order.shape[0]
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        img = int(image_ids[order[d]])
        bb = bb_all[order[d]]
        bbgt = np.asarray(gt_boxes.get(img, np.zeros((0, 4))), np.float64).reshape(-1, 4)
        ovmax, jmax = -np.inf, -1
        if bbgt.size > 0:
            iw = np.maximum(np.minimum(bbgt[:, 2], bb[2]) - np.maximum(bbgt[:, 0], bb[0]), 0.)
            ih = np.maximum(np.minimum(bbgt[:, 3], bb[3]) - np.maximum(bbgt[:, 1], bb[1]), 0.)
            inters = iw * ih
            uni = (bb[2] - bb[0]) * (bb[3] - bb[1]) + (bbgt[:, 2] - bbgt[:, 0]) * (bbgt[:, 3] - bbgt[:, 1]) - inters
            overlaps = inters / uni
            ovmax, jmax = np.max(overlaps), int(np.argmax(overlaps))
        if ovmax > ovthresh:
            if not np.asarray(gt_difficult[img]).astype(bool)[jmax]:
                if not det[img][jmax]:
                    tp[d] = 1.
                    det[img][jmax] = True
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    ap = average_precision_voc07(prec, rec) if use_07_metric else average_precision_voc12(prec, rec)
    return rec, prec, ap


def evaluate(stream, voc07=True, voc12=True):
    """eval_ron_network.py:289-335: per-class AP and mAP from a StreamingTpFp.  Returns dict name -> value with the
    reference's summary names ('AP_VOC07/<c>', 'AP_VOC07/mAP', ...)."""
    out = {}
    aps07, aps12 = [], []
    for c in stream.labels:
        prec, rec = precision_recall(*stream.arrays(c))
        if voc07:
            v = average_precision_voc07(prec, rec)
            out['AP_VOC07/%s' % c] = v
            aps07.append(v)
        if voc12:
            v = average_precision_voc12(prec, rec)
            out['AP_VOC12/%s' % c] = v
            aps12.append(v)
    if voc07:
        out['AP_VOC07/mAP'] = float(np.mean(aps07))
    if voc12:
        out['AP_VOC12/mAP'] = float(np.mean(aps12))
    return out


def detection_agreement(got, ref, tol=1e-4):
    """How well one detection list reproduces another ("detection agreement rate" on synthetic inputs,
    the stand-in for mAP while no checkpoint ships).  `got` / `ref`: dicts with classes, anchor_index, scores, bboxes
    (one image).  A reference detection is *reproduced* when `got` holds the same (class, anchor_index) pair.

    Returns dict(n_ref, n_got, reproduced, within_tol, max_score_diff, max_box_diff):
      reproduced  fraction of the reference detections present in `got`;
      within_tol  fraction of the reproduced ones whose score AND box agree within `tol` (north-star: 1e-4);
      max_*_diff  largest absolute differences over the reproduced ones."""
    ref_keys = {(int(c), int(a)): k for k, (c, a) in enumerate(zip(ref['classes'], ref['anchor_index']))}
    hit = [(k, ref_keys[(int(c), int(a))]) for k, (c, a) in enumerate(zip(got['classes'], got['anchor_index']))
           if (int(c), int(a)) in ref_keys]
    out = dict(n_ref=len(ref_keys), n_got=int(len(got['classes'])), reproduced=1.0, within_tol=1.0,
               max_score_diff=0.0, max_box_diff=0.0)
    if ref_keys:
        out['reproduced'] = len(hit) / float(len(ref_keys))
    if hit:
        gi, ri = np.array([h[0] for h in hit]), np.array([h[1] for h in hit])
        ds = np.abs(np.asarray(got['scores'])[gi] - np.asarray(ref['scores'])[ri])
        db = np.abs(np.asarray(got['bboxes'])[gi] - np.asarray(ref['bboxes'])[ri]).max(axis=1)
        out['within_tol'] = float(np.mean((ds <= tol) & (db <= tol)))
        out['max_score_diff'], out['max_box_diff'] = float(ds.max()), float(db.max())
    return out
