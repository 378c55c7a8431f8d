"""Host-side wrappers of the post-processing and single-operator entry points.

Each function takes/returns torch tensors on the GPU (device containers only) and calls one
C-ABI entry point of libron_hip.so.  Names follow the reference's numpy module
(``nets/np_methods.py``) where a function replaces one of its steps.
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Detections, Heads, PostCfg, check, current_stream, lib, ptr


# --------------------------------------------------------------------------- #
# anchors (host)                       replaces nets/ron_vgg_320.py:285-355
# --------------------------------------------------------------------------- #
def anchor_one_layer(img_shape, feat_shape, sizes, ratios, step, offset=0.5, dtype=np.float32):
    """(y, x, h, w) with the reference's shapes: y, x [H, W, 1]; h, w [A]."""
    fh, fw = int(feat_shape[0]), int(feat_shape[1])
    na = len(sizes) * len(ratios)
    y = np.empty((fh, fw, 1), np.float32)
    x = np.empty((fh, fw, 1), np.float32)
    h = np.empty((na,), np.float32)
    w = np.empty((na,), np.float32)
    sz = (C.c_double * len(sizes))(*[float(s) for s in sizes])
    rt = (C.c_double * len(ratios))(*[float(r) for r in ratios])
    check(lib().ron_anchor_one_layer(int(img_shape[0]), int(img_shape[1]), fh, fw, sz, len(sizes), rt, len(ratios),
                                     float(step), float(offset), ptr(y), ptr(x), ptr(h), ptr(w)))
    return y.astype(dtype, copy=False), x.astype(dtype, copy=False), h.astype(dtype, copy=False), w.astype(dtype, copy=False)


def anchors_to_device(anchors, device):
    """List of (y, x, h, w) numpy -> list of 4-tuples of flat float32 device tensors."""
    out = []
    for (y, x, h, w) in anchors:
        out.append(tuple(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(-1)).to(device)
                         for a in (y, x, h, w)))
    return out


# --------------------------------------------------------------------------- #
# detection records
# --------------------------------------------------------------------------- #
class DetectionBuffers(object):
    """Fixed-capacity per-image detection lists (SURVEY.md 8b "detection record")."""
    FIELDS = ('classes', 'scores', 'bboxes', 'anchor_index', 'count')

    def __init__(self, n, capacity, device):
        self.n, self.capacity = n, capacity
        self.classes = torch.zeros((n, capacity), dtype=torch.int32, device=device)
        self.scores = torch.zeros((n, capacity), dtype=torch.float32, device=device)
        self.bboxes = torch.zeros((n, capacity, 4), dtype=torch.float32, device=device)
        self.anchor_index = torch.zeros((n, capacity), dtype=torch.int32, device=device)
        self.count = torch.zeros((n,), dtype=torch.int32, device=device)

    def narrow(self, n):
        """The first `n` images of this set as a DetectionBuffers over the same memory (leading-dimension views)."""
        assert 1 <= n <= self.n
        if n == self.n:
            return self
        v = object.__new__(DetectionBuffers)
        v.n, v.capacity = n, self.capacity
        for f in self.FIELDS:
            setattr(v, f, getattr(self, f)[:n])
        return v

    def record_stream(self, stream):
        """Tell the caching allocator that `stream` uses these tensors (they were allocated on another stream)."""
        for f in self.FIELDS:
            getattr(self, f).record_stream(stream)

    def c_struct(self):
        return Detections(self.capacity, ptr(self.classes), ptr(self.scores), ptr(self.bboxes),
                          ptr(self.anchor_index), ptr(self.count))

    def to_lists(self):
        """Host copy: list (per image) of dicts with the `count` valid rows."""
        cnt = self.count.cpu().numpy()
        cl, sc, bb, ai = (t.cpu().numpy() for t in (self.classes, self.scores, self.bboxes, self.anchor_index))
        return [dict(classes=cl[i, :cnt[i]].astype(np.int64), scores=sc[i, :cnt[i]], bboxes=bb[i, :cnt[i]],
                     anchor_index=ai[i, :cnt[i]].astype(np.int64)) for i in range(self.n)]


def _fill_heads(cls, obj, loc, anchors_dev, num_classes):
    h = Heads()
    h.num_layers = len(cls)
    h.num_classes = num_classes
    keep = []
    for i in range(len(cls)):
        c = cls[i]
        assert c.dtype == torch.float32 and c.is_cuda and c.dim() == 5, 'cls[%d] must be a float32 GPU tensor [N,H,W,A,C]' % i
        h.feat_h[i], h.feat_w[i], h.num_anchors[i] = c.shape[1], c.shape[2], c.shape[3]
        tensors = [c.contiguous(), None if obj is None else obj[i].contiguous(), loc[i].contiguous()]
        keep.append(tensors)
        h.cls[i] = tensors[0].data_ptr()
        h.obj[i] = None if tensors[1] is None else tensors[1].data_ptr()
        h.loc[i] = tensors[2].data_ptr()
        if anchors_dev is not None:
            ay, ax, ah, aw = anchors_dev[i]
            h.anchor_y[i], h.anchor_x[i] = ay.data_ptr(), ax.data_ptr()
            h.anchor_h[i], h.anchor_w[i] = ah.data_ptr(), aw.data_ptr()
    return h, keep


_WORKSPACES = collections.OrderedDict()
_MAX_WORKSPACES = 8


def _workspace(device, nbytes):
    """Scratch of the post-processing entry points, one per (device, stream): calls on different streams never share
    candidate lists, and a regrown buffer is dropped on the stream that was its only user.  At most _MAX_WORKSPACES
    entries, least recently used first out, so short-lived streams do not pin scratch for the life of the process; an
    entry is (buffer, stream object): holding the stream keeps its handle from being recycled for another stream while
    the allocator still attributes the buffer to it."""
    stream = torch.cuda.current_stream(device)
    key = (device.type, device.index, stream.cuda_stream)
    ent = _WORKSPACES.get(key)
    if ent is None or ent[0].numel() < nbytes:
        ent = (torch.empty((int(nbytes),), dtype=torch.uint8, device=device), stream)
        _WORKSPACES[key] = ent
    _WORKSPACES.move_to_end(key)
    while len(_WORKSPACES) > _MAX_WORKSPACES:
        _WORKSPACES.popitem(last=False)
    return ent[0]


def post_np(cls, obj, loc, anchors_dev, num_classes=None, objectness_thres=0.03, select_threshold=0.01,
            nms_threshold=0.45, top_k=400, bbox_img=(0., 0., 1., 1.), prior_scaling=(0.1, 0.1, 0.2, 0.2),
            cls_is_prob=False, obj_is_prob=False, loc_decoded=False, want_sorted=False):
    """np_methods pipeline on the GPU (ron_post_np): per-layer lists of GPU tensors in, DetectionBuffers out.

    select -> clip -> sort(top_k) -> class-aware IoU NMS -> resize, nets/np_methods.py:56-242, with the softmax
    and the objectness gate of eval_ron_network.py:227-229 in front when logits are given.
    Returns (detections, sorted_or_None, n_candidates[int32 N]).
    """
    n = cls[0].shape[0]
    dev = cls[0].device
    if num_classes is None:
        num_classes = int(cls[0].shape[-1])          # RONParams.num_classes = the class tensors' last axis
    heads, keep = _fill_heads(cls, obj, loc, None if loc_decoded else anchors_dev, num_classes)
    cfg = PostCfg()
    # select_threshold None / 0: the arg-max branch of ssd_bboxes_select_layer (np_methods.py:82-89)
    cfg.objectness_thres, cfg.select_threshold, cfg.nms_threshold = objectness_thres, (select_threshold or 0.0), nms_threshold
    cfg.top_k = top_k
    for i in range(4):
        cfg.bbox_img[i] = bbox_img[i]
        cfg.prior_scaling[i] = prior_scaling[i]
    cfg.input_flags = ((_lib.RON_IN_CLS_IS_PROB if cls_is_prob else 0) | (_lib.RON_IN_OBJ_IS_PROB if obj_is_prob else 0) |
                       (_lib.RON_IN_LOC_DECODED if loc_decoded else 0))
    nbytes = lib().ron_post_np_workspace_bytes(C.byref(heads), n)
    if nbytes < 0:
        check(-1)
    ws = _workspace(dev, nbytes)
    out = DetectionBuffers(n, top_k, dev)
    srt = DetectionBuffers(n, top_k, dev) if want_sorted else None
    n_cand = torch.zeros((n,), dtype=torch.int32, device=dev)
    out_c = out.c_struct()
    srt_c = srt.c_struct() if srt is not None else None
    check(lib().ron_post_np(C.byref(heads), n, C.byref(cfg), ptr(ws), nbytes, C.byref(out_c),
                            C.byref(srt_c) if srt_c is not None else None, ptr(n_cand), current_stream()))
    del keep
    return out, srt, n_cand


def np_sort_nms(classes, scores, bboxes, top_k=400, nms_threshold=0.45, n_valid=None, want_sorted=False):
    """bboxes_sort -> bboxes_nms (np_methods.py:137-150, :229-242) on explicit lists [N, K] / [N, K, 4]."""
    n, n_in = scores.shape
    dev = scores.device
    classes = classes.to(torch.int32).contiguous()
    scores = scores.contiguous()
    bboxes = bboxes.contiguous()
    nbytes = lib().ron_np_sort_nms_workspace_bytes(n, n_in)
    ws = _workspace(dev, nbytes)
    out = DetectionBuffers(n, top_k, dev)
    srt = DetectionBuffers(n, top_k, dev) if want_sorted else None
    out_c = out.c_struct()
    srt_c = srt.c_struct() if srt is not None else None
    check(lib().ron_np_sort_nms(ptr(classes), ptr(scores), ptr(bboxes), ptr(n_valid), n, n_in, top_k,
                                float(nms_threshold), ptr(ws), nbytes, C.byref(out_c),
                                C.byref(srt_c) if srt_c is not None else None, current_stream()))
    return out, srt


def bboxes_decode_layer(loc, anchor_dev, prior_scaling=(0.1, 0.1, 0.2, 0.2)):
    """ssd_bboxes_decode for one layer (np_methods.py:23-53 == ssd_common.py:448-474), batch capable."""
    loc = loc.contiguous()
    n, fh, fw, a, _ = loc.shape
    out = torch.empty_like(loc)
    ps = (C.c_float * 4)(*prior_scaling)
    ay, ax, ah, aw = anchor_dev
    check(lib().ron_bboxes_decode_layer(ptr(loc), n, fh, fw, a, ptr(ay), ptr(ax), ptr(ah), ptr(aw), ps, ptr(out),
                                        current_stream()))
    return out


def bboxes_filter_min(scores, bboxes, top_k, minsize=0.03):
    """RONNet.bboxes_filter_min on tensors (nets/ron_vgg_320.py:217-233): scores [B, N], bboxes [B, N, 4] -> per list the rows with
    w > minsize and h > minsize in their order (tf.boolean_mask), zero padded to top_k rows - or to the longest list's count when that
    is larger (tfe_tensors.pad_axis only ever pads).  The reference squeezes axis 0, i.e. takes B = 1; any B works here.
    The output length depends on the data, so this call reads the counts back (one host synchronisation), like a TF session run."""
    scores = scores.to(torch.float32).contiguous()
    bboxes = bboxes.to(torch.float32).contiguous()
    assert scores.dim() == 2 and bboxes.shape == scores.shape + (4,), 'scores [B, N], bboxes [B, N, 4]'
    b, n = scores.shape
    if n == 0:                                  # nothing to filter: top_k rows of padding (pad_axis)
        return (torch.zeros((b, int(top_k)), dtype=torch.float32, device=scores.device),
                torch.zeros((b, int(top_k), 4), dtype=torch.float32, device=scores.device))
    rows = max(n, int(top_k))
    out_s = torch.empty((b, rows), dtype=torch.float32, device=scores.device)
    out_b = torch.empty((b, rows, 4), dtype=torch.float32, device=scores.device)
    counts = torch.empty((b,), dtype=torch.int32, device=scores.device)
    check(lib().ron_bboxes_filter_min(ptr(scores), ptr(bboxes), b, n, float(minsize), ptr(out_s), ptr(out_b), rows, ptr(counts), current_stream()))
    keep = max(int(counts.max().item()), int(top_k))
    return out_s[:, :keep], out_b[:, :keep]


def softmax_last(x, pick=-1):
    """slim.softmax over the last axis; pick >= 0 keeps only that channel (shape [..., 1])."""
    x = x.contiguous()
    c = x.shape[-1]
    rows = x.numel() // c
    out = torch.empty(x.shape if pick < 0 else x.shape[:-1] + (1,), dtype=torch.float32, device=x.device)
    check(lib().ron_softmax_last(ptr(x), rows, c, pick, ptr(out), current_stream()))
    return out


# --------------------------------------------------------------------------- #
# single operators (parity tests of the conv kernels)
# --------------------------------------------------------------------------- #
def conv2d_nhwc(x, w, bias=None, residual=None, stride=1, dilation=1, relu=True, transpose=False, dtype='bf16',
                tile_cfg=-1, splitk=-1, pool=False, in_cstride=0, in_coff=0, center_from=0):
    """x GPU fp32 [N,H,W,Cin]; w, bias host numpy (HWIO, or [kh,kw,Cout,Cin] when transpose)."""
    x = x.contiguous()
    w = np.ascontiguousarray(w, dtype=np.float32)
    n, h, wd, cin = x.shape
    kh, kw = w.shape[:2]
    cout = w.shape[2] if transpose else w.shape[3]
    d = _lib.ConvDesc(n, h, wd, cin, cout, kh, kw, stride, dilation, int(relu), int(transpose), _lib.DTYPES[dtype], tile_cfg, in_cstride, in_coff, int(pool), splitk, center_from)
    ho, wo = (h * stride, wd * stride) if transpose else (h // stride, wd // stride)
    if pool:
        ho, wo = ho // 2, wo // 2
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    r = None if residual is None else residual.contiguous()
    check(lib().ron_conv2d_nhwc(C.byref(d), ptr(x), ptr(w), ptr(b), ptr(r), ptr(y), current_stream()))
    return y


def conv_plan(n, h, w, cin, cout, k=3, stride=1, dilation=1, dtype='bf16', tile_cfg=-1, splitk=-1, pool=False, center_from=0, transpose=False):
    """How conv2d_nhwc would run this convolution (ron_conv_plan): dict(tile_cfg, splitk, tile_order, taps_inner)."""
    d = _lib.ConvDesc(n, h, w, cin, cout, k, k, stride, dilation, 1, int(transpose), _lib.DTYPES[dtype], tile_cfg, 0, 0, int(pool), splitk, center_from)
    out = (C.c_int32 * 4)()
    check(lib().ron_conv_plan(C.byref(d), out))
    return dict(tile_cfg=out[0], splitk=out[1], tile_order=out[2], taps_inner=out[3])


def conv2d_heads_nhwc(x, w, split_first, bias=None, dilation=1, relu=False, dtype='bf16', tile_cfg=-1, splitk=-1):
    """One convolution, two fp32 head tensors (ron_conv2d_heads_nhwc: what the SSD-512 graph does with the class and box convolutions of
    a feature layer, nets/ssd_vgg_300.py:403-431): w HWIO with cout = both heads' channels, the first `split_first` of them -> y_first."""
    x = x.contiguous()
    w = np.ascontiguousarray(w, dtype=np.float32)
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    d = _lib.ConvDesc(n, h, wd, cin, cout, kh, kw, 1, dilation, int(relu), 0, _lib.DTYPES[dtype], tile_cfg, 0, 0, 0, splitk, 0)
    y1 = torch.empty((n, h, wd, split_first), dtype=torch.float32, device=x.device)
    y2 = torch.empty((n, h, wd, cout - split_first), dtype=torch.float32, device=x.device)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    check(lib().ron_conv2d_heads_nhwc(C.byref(d), int(split_first), ptr(x), ptr(w), ptr(b), ptr(y1), ptr(y2), current_stream()))
    return y1, y2


def maxpool2x2_nhwc(x, dtype='bf16'):
    x = x.contiguous()
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    check(lib().ron_maxpool2x2_nhwc(ptr(x), n, h, w, c, _lib.DTYPES[dtype], ptr(y), current_stream()))
    return y
