"""Post-processing of the reference's second evaluation harness, ``ron_eval.py``, on the GPU:
``flaten_predict`` (ron_eval.py:111-144) -> ``tfe.bboxes_clip`` -> ``filter_boxes`` (:369-392) -> ``tf_bboxes_nms``
(:146-206, all classes together) -> ``tfe.bboxes_resize`` (:466-477), one ``ron_post_eval`` call for a batch.

Unlike the np_methods / detected_bboxes pipelines the score here is objectness x class probability, every anchor gets ONE
label (argmax over all classes, background included) and the NMS is class agnostic with overlap 'union'."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import EvalCfg, check, current_stream, lib, ptr
from .ops import DetectionBuffers, _fill_heads, _workspace

NMS_MODES = {'min': 0, 'union': 1}


def filter_min_size(image_shapes, net_input_shape=(320., 320.), min_size_ratio=0.03):
    """min_size of ``filter_boxes`` (ron_eval.py:373-374), float32 like the TF graph: per image [h, w] of the ORIGINAL image."""
    hw = np.asarray(image_shapes, np.float32).reshape(-1, 2)
    area = (hw[:, 0].astype(np.int64) * hw[:, 1].astype(np.int64)).astype(np.float32)
    den = np.float32(net_input_shape[0] * net_input_shape[1])
    return np.maximum(np.float32(0.0001), np.float32(min_size_ratio) * np.sqrt(area / den)).astype(np.float32)


def post_eval(cls, obj, loc, anchors_dev, image_shapes, num_classes=21, objectness_thres=0.95, select_threshold=0.6,
              nms_threshold=0.4, keep_top_k=20, nms_mode='union', bbox_img=(0., 0., 1., 1.), min_size_ratio=0.03,
              net_input_shape=(320., 320.), prior_scaling=(0.1, 0.1, 0.2, 0.2), cls_is_prob=True, obj_is_prob=True,
              loc_decoded=True, nms_by_class=False):
    """Per-layer lists of GPU tensors in (the outputs of ``RONNet.net`` + ``bboxes_decode`` by default), DetectionBuffers
    out: ``classes`` = labels, rows in score order, ``count`` <= keep_top_k.  Defaults are ron_eval.py's flags (:82-92).
    ``nms_by_class``: True or 'v1' = ``tf_bboxes_nms_by_class_v1`` (ron_eval.py:282-366) instead of ``tf_bboxes_nms``: a kept box
    only suppresses boxes of its own label.  'scores' = ``tf_bboxes_nms_by_class`` (ron_eval.py:212-280, the function the commented call
    of :474 names): every score column (objectness x class probability, background included) runs its own greedy NMS with at most
    ``keep_top_k`` picks; a row some column kept comes back with the largest of its kept scores and that column as its class, rows in
    the flattened ANCHOR order, up to num_classes * keep_top_k of them (the buffers get that capacity)."""
    if nms_by_class not in (False, True, 'v1', 'scores'):
        raise ValueError("nms_by_class must be False, True / 'v1' or 'scores'")
    per_column = nms_by_class == 'scores'
    if nms_mode not in NMS_MODES:
        raise ValueError('unknown mode to use for nms.')          # ron_eval.py:188
    n, dev = cls[0].shape[0], cls[0].device
    heads, keep = _fill_heads(cls, obj, loc, None if loc_decoded else anchors_dev, num_classes)
    cfg = EvalCfg()
    cfg.objectness_thres, cfg.select_threshold, cfg.nms_threshold = objectness_thres, select_threshold, nms_threshold
    cfg.keep_top_k, cfg.nms_mode = keep_top_k, NMS_MODES[nms_mode] | (4 if per_column else 2 if nms_by_class else 0)
    for i in range(4):
        cfg.bbox_img[i] = bbox_img[i]
        cfg.prior_scaling[i] = prior_scaling[i]
    cfg.input_flags = ((_lib.RON_IN_CLS_IS_PROB if cls_is_prob else 0) | (_lib.RON_IN_OBJ_IS_PROB if obj_is_prob else 0) |
                       (_lib.RON_IN_LOC_DECODED if loc_decoded else 0))
    ms = torch.from_numpy(filter_min_size(image_shapes, net_input_shape, min_size_ratio)).to(dev)
    assert ms.shape[0] == n, 'one (height, width) per image'
    nbytes = lib().ron_post_eval_workspace_bytes_mode(C.byref(heads), n, cfg.nms_mode)
    if nbytes < 0:
        check(-1)
    ws = _workspace(dev, nbytes)
    out = DetectionBuffers(n, num_classes * keep_top_k if per_column else keep_top_k, dev)
    oc = out.c_struct()
    check(lib().ron_post_eval(C.byref(heads), n, ptr(ms), C.byref(cfg), ptr(ws), nbytes, C.byref(oc), current_stream()))
    del keep
    return out
