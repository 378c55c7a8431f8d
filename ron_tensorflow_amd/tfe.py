"""TF-evaluation post-processing variant on the GPU (ron_post_tfe): what RONNet.detected_bboxes computes in the
reference (nets/ron_vgg_320.py:234-256, tf_extended/bboxes.py:60-302), returned in the reference's form:
two dicts  class -> scores [N, keep_top_k]  /  class -> bboxes [N, keep_top_k, 4], zero padded."""
import ctypes as C

import torch

from . import _lib
from ._lib import TfeCfg, check, current_stream, lib, ptr
from .ops import _fill_heads, _workspace

NMS_MODES = {'min': 0, 'union': 1}


def post_tfe(cls, obj, loc, anchors_dev, num_classes=21, objectness_thres=0.03, select_threshold=None,
             nms_threshold=0.5, clipping_bbox=None, top_k=400, keep_top_k=200, nms_mode='min', min_size=0.03,
             prior_scaling=(0.1, 0.1, 0.2, 0.2), cls_is_prob=True, obj_is_prob=True, loc_decoded=True):
    """Dense outputs: scores [N, C-1, keep_top_k], bboxes [N, C-1, keep_top_k, 4]."""
    if nms_mode not in NMS_MODES:
        raise ValueError('unknown mode to use for nms.')          # tf_extended/bboxes.py:210
    n, dev = cls[0].shape[0], cls[0].device
    heads, keep = _fill_heads(cls, obj, loc, None if loc_decoded else anchors_dev, num_classes)
    cfg = TfeCfg()
    cfg.objectness_thres = objectness_thres
    cfg.select_threshold = 0.0 if select_threshold is None else select_threshold     # ssd_common.py:521
    cfg.nms_threshold, cfg.top_k, cfg.keep_top_k, cfg.nms_mode = nms_threshold, top_k, keep_top_k, NMS_MODES[nms_mode]
    cfg.clip = 0 if clipping_bbox is None else 1
    cfg.min_size = -1.0 if min_size is None else min_size
    for i in range(4):
        cfg.clipping_bbox[i] = 0.0 if clipping_bbox is None else clipping_bbox[i]
        cfg.prior_scaling[i] = prior_scaling[i]
    cfg.input_flags = ((_lib.RON_IN_CLS_IS_PROB if cls_is_prob else 0) | (_lib.RON_IN_OBJ_IS_PROB if obj_is_prob else 0) |
                       (_lib.RON_IN_LOC_DECODED if loc_decoded else 0))
    nbytes = lib().ron_post_tfe_workspace_bytes(C.byref(heads), n)
    if nbytes < 0:
        check(-1)
    ws = _workspace(dev, nbytes)
    scores = torch.empty((n, num_classes - 1, keep_top_k), dtype=torch.float32, device=dev)
    bboxes = torch.empty((n, num_classes - 1, keep_top_k, 4), dtype=torch.float32, device=dev)
    check(lib().ron_post_tfe(C.byref(heads), n, C.byref(cfg), ptr(ws), nbytes, ptr(scores), ptr(bboxes), current_stream()))
    del keep
    return scores, bboxes


def detected_bboxes(predictions, localisations, num_classes=21, select_threshold=None, nms_threshold=0.5,
                    clipping_bbox=None, top_k=400, keep_top_k=200, nms_mode='min', min_size=0.03):
    """Reference signature (RONNet.detected_bboxes): `predictions` are the (already objectness-gated) class
    probabilities, `localisations` the decoded boxes; returns (dict_scores, dict_bboxes)."""
    scores, bboxes = post_tfe(predictions, None, localisations, None, num_classes=num_classes,
                              select_threshold=select_threshold, nms_threshold=nms_threshold, clipping_bbox=clipping_bbox,
                              top_k=top_k, keep_top_k=keep_top_k, nms_mode=nms_mode, min_size=min_size)
    return ({c: scores[:, c - 1] for c in range(1, num_classes)},
            {c: bboxes[:, c - 1] for c in range(1, num_classes)})
