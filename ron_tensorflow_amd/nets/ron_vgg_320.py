"""RON-320 (VGG-16) network object with the interface of the reference's ``nets/ron_vgg_320.py``.

``RONNet`` keeps the names, argument names, defaults and return arity/order of the reference class
(nets/ron_vgg_320.py:86-279) so that an ``eval_ron_network.py``-style driver works unchanged, but every
tensor op runs as HIP kernels through libron_hip.so; tensors are torch CUDA tensors (device containers).

Additions: ``variant`` ('reducedfc' = what the reference's RONNet.net builds, :144; 'full' = ron_net,
reachable in the reference through nets_factory.get_network_fn), ``dtype`` ('bf16' | 'fp16' | 'fp32' | 'f16x3' = split precision: fp32-grade results on the f16 matrix cores),
``load_weights`` (dict keyed by TF variable names) and the fused ``detect``.
"""
import contextlib
import ctypes as C
from collections import namedtuple

import numpy as np
import torch

from .. import _lib, ops
from .._lib import check, current_stream, lib, ptr

# same field names as the reference namedtuple (nets/ron_vgg_320.py:72-83)
RONParams = namedtuple('SSDParameters', ['img_shape', 'num_classes', 'no_annotation_label', 'feat_layers',
                                         'feat_shapes', 'allowed_borders', 'anchor_sizes', 'anchor_ratios',
                                         'anchor_steps', 'anchor_offset', 'prior_scaling'])


class _DataFormatScope(object):
    """What `arg_scope(data_format=...)` returns: inside the `with`, the networks it was made for take / return that layout."""

    def __init__(self, nets, data_format):
        if data_format not in ('NHWC', 'NCHW'):
            raise ValueError('data_format must be NHWC or NCHW, not %r' % (data_format,))
        self.nets, self.data_format, self._saved = nets, data_format, []

    def __enter__(self):
        self._saved = [getattr(n, '_data_format', 'NHWC') for n in self.nets]
        for n in self.nets:
            n._data_format = self.data_format
        return self

    def __exit__(self, *exc):
        for n, f in zip(self.nets, self._saved):
            n._data_format = f
        return False


class RONNet(object):
    """RON VGG-based 320 network: conv4 -> 40x40, conv5 -> 20x20, fc6 -> 10x10, fc7 -> 5x5."""
    default_params = RONParams(
        img_shape=(320, 320),
        num_classes=21,
        no_annotation_label=21,
        feat_layers=['block7', 'block6', 'block5', 'block4'],
        feat_shapes=[(5, 5), (10, 10), (20, 20), (40, 40)],
        allowed_borders=[32, 16, 8, 4],
        anchor_sizes=[(224., 256.), (160., 192.), (96., 128.), (32., 64.)],
        anchor_ratios=[[1, 2, 3, 1. / 2, 1. / 3]] * 4,
        anchor_steps=[64, 32, 16, 8],
        anchor_offset=0.5,
        prior_scaling=[0.1, 0.1, 0.2, 0.2])

    def __init__(self, params=None, variant='reducedfc', dtype='bf16', max_batch=32, device=None, fuse_pools=False,
                 multi_stream=False, group_heads=True, head_plan=None):
        self.params = params if isinstance(params, RONParams) else RONNet.default_params
        if variant not in _lib.VARIANTS:
            raise ValueError('Unknown RON variant %s' % variant)
        if dtype not in _lib.DTYPES:
            raise ValueError('Unknown dtype %s' % dtype)
        self.variant, self.dtype, self.max_batch = variant, dtype, max_batch
        # fuse_pools: block1..block3 are never written at full resolution (their max-pool runs in the conv epilogue);
        # end_points then offers block4..block7 only
        self.fuse_pools = fuse_pools
        # multi_stream: heads of block7/6/5 on side streams (fork/join inside every forward)
        self.multi_stream = multi_stream
        # group_heads: the small independent head convolutions of the coarse scales share launches (RON_CFG_NO_GROUPS off)
        self.no_groups = not group_heads
        # head_plan: None = by max_batch (<= 4: one launch per dependency level, else the batch plan); 'level' / 'batch' force one
        if head_plan not in (None, 'level', 'batch'):
            raise ValueError('head_plan must be None, "level" or "batch"')
        self.head_plan = head_plan
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self._ctx = None
        self._anchors_dev = None

    # ------------------------------------------------------------------ context / weights
    def _context(self):
        if self._ctx is None:
            cfg = _lib.Config(_lib.VARIANTS[self.variant], _lib.DTYPES[self.dtype], self.params.img_shape[0],
                              self.params.img_shape[1], self.params.num_classes, self.max_batch,
                              self.device.index or 0,
                              (_lib.RON_CFG_FUSE_POOLS if self.fuse_pools else 0) |
                              (_lib.RON_CFG_MULTI_STREAM if getattr(self, 'multi_stream', False) else 0) |
                              (_lib.RON_CFG_NO_STEM2 if getattr(self, 'no_stem2', False) else 0) |
                              (_lib.RON_CFG_NO_GROUPS if getattr(self, 'no_groups', False) else 0) |
                              (_lib.RON_CFG_NO_HALO_SKIP if getattr(self, 'no_halo_skip', False) else 0) |
                              {None: 0, 'level': _lib.RON_CFG_LEVEL_GROUPS, 'batch': _lib.RON_CFG_BATCH_GROUPS}[getattr(self, 'head_plan', None)])
            h = C.c_void_p()
            check(lib().ron_create(C.byref(h), C.byref(cfg)))
            self._ctx = h
        return self._ctx

    def variables(self):
        """[(tf_name, shape)] the graph expects (ron_variable_info)."""
        ctx = self._context()
        out = []
        for i in range(lib().ron_num_variables(ctx)):
            name = C.c_char_p()
            shape = (C.c_int64 * 4)()
            nd = C.c_int()
            check(lib().ron_variable_info(ctx, i, C.byref(name), shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[k] for k in range(nd.value))))
        return out

    def load_weights(self, weights):
        """weights: dict {tf variable name: ndarray} (HWIO convs, [kh,kw,Cout,Cin] transposed convs)."""
        ctx = self._context()
        for name, shape in self.variables():
            if name not in weights:
                raise KeyError('missing variable %s' % name)
            a = np.ascontiguousarray(weights[name], dtype=np.float32)
            shp = (C.c_int64 * 4)(*a.shape)
            check(lib().ron_load_weight(ctx, name.encode(), ptr(a), shp, a.ndim))
        check(lib().ron_finalize_weights(ctx))
        return self

    def load_checkpoint(self, checkpoint_path, checkpoint_model_scope=None, checkpoint_exclude_scopes=None,
                        ignore_missing_vars=False, extra=None):
        """Restore from a TensorFlow V2 checkpoint (prefix or directory) with the reference's rules
        (tf_utils.py:184-243; eval_ron_network.py:346-361), or from an ``.npz`` of ``weights.save_npz``.
        Every variable of the graph is needed to run it: excluded / missing ones must come from ``extra``
        (dict name -> ndarray), otherwise ``load_weights`` raises KeyError."""
        from .. import checkpoint, weights as W
        if str(checkpoint_path).endswith('.npz'):
            found = W.load_npz(checkpoint_path)
        else:
            model_name = self.variables()[0][0].split('/')[0]
            found = checkpoint.load_checkpoint(checkpoint_path, self.variables(), model_name=model_name,
                                               checkpoint_model_scope=checkpoint_model_scope,
                                               checkpoint_exclude_scopes=checkpoint_exclude_scopes,
                                               ignore_missing_vars=ignore_missing_vars)
        if extra:
            found = dict(extra, **found)
        return self.load_weights(found)

    def launch_plan(self):
        """Names of the launches of one forward pass in order (ron_profile_get): a grouped launch reads 'group[first+N]', its other
        members '(name)'; the last entry is the post-processing stage."""
        import ctypes as C
        ctx, names = self._context(), []
        for i in range(lib().ron_profile_num_ops(ctx)):
            name = C.c_char_p()
            check(lib().ron_profile_get(ctx, i, C.byref(name), None, None, None, None, None, None))
            names.append(name.value.decode())
        return names

    def grouped_launches(self):
        return int(lib().ron_num_grouped_launches(self._context()))

    def flops_per_image(self):
        return lib().ron_flops_per_image(self._context())

    def clone(self):
        """A second execution slot over the same device weights (ron_clone): its own activations, scratch and
        streams, so that another batch can be in flight on another stream.  The slot keeps this network alive."""
        import copy
        other = copy.copy(self)
        h = C.c_void_p()
        check(lib().ron_clone(self._context(), C.byref(h)))
        other._ctx = h
        other._weights_from = self
        other._slots = []
        self._slots = getattr(self, '_slots', [])
        self._slots.append(other)
        return other

    def close(self):
        for slot in getattr(self, '_slots', []):       # slots borrow this network's weights: they go first
            slot.close()
        self._slots = []
        if self._ctx is not None:
            check(lib().ron_destroy(self._ctx))
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference interface
    def _head_geometry(self):
        """[(feat_h, feat_w, anchors per cell)] per scale and the class count, as the context computes them from the image size
        (ron_heads_describe) -- NOT from params.feat_shapes, which callers may replace (SSDNet.update_feature_shapes does)."""
        if getattr(self, '_geom', None) is None:
            hd = _lib.Heads()
            check(lib().ron_heads_describe(self._context(), C.byref(hd)))
            self._geom = ([(int(hd.feat_h[i]), int(hd.feat_w[i]), int(hd.num_anchors[i])) for i in range(hd.num_layers)],
                          int(hd.num_classes))
        return self._geom

    def _head_buffers(self, n):
        geom, nc = self._head_geometry()
        cls, obj, loc = [], [], []
        for (fh, fw, A) in geom:
            cls.append(torch.empty((n, fh, fw, A, nc), dtype=torch.float32, device=self.device))
            obj.append(torch.empty((n, fh, fw, A, 2), dtype=torch.float32, device=self.device))
            loc.append(torch.empty((n, fh, fw, A, 4), dtype=torch.float32, device=self.device))
        return cls, obj, loc

    def forward_heads(self, inputs):
        """Conv stack only: (logits, objness_logits, localisations) as lists of fp32 GPU tensors."""
        inputs = inputs.to(self.device, torch.float32).contiguous()
        n = inputs.shape[0]
        assert tuple(inputs.shape[1:]) == tuple(self.params.img_shape) + (3,), 'inputs must be [N,%d,%d,3] NHWC' % self.params.img_shape
        cls, obj, loc = self._head_buffers(n)
        hd = _lib.Heads()
        for i in range(4):
            hd.cls[i], hd.obj[i], hd.loc[i] = cls[i].data_ptr(), obj[i].data_ptr(), loc[i].data_ptr()
        check(lib().ron_forward(self._context(), ptr(inputs), n, C.byref(hd), current_stream()))
        return cls, obj, loc

    def net(self, inputs, is_training=True, dropout_keep_prob=0.5, prediction_fn=None, reuse=None,
            scope='ron_320_vgg', end_points=('block1', 'block2', 'block3', 'block4', 'block5', 'block6', 'block7')):
        """RON network (nets/ron_vgg_320.py:136-154).  Inference only: `is_training` / `dropout_keep_prob` /
        `reuse` / `scope` are accepted for signature compatibility (the reference's dropout is commented out,
        :480, and BatchNorm runs on its moving statistics).  Returns the reference's 6-tuple
        (predictions, logits, objness_pred, objness_logits, localisations, end_points)."""
        nchw = getattr(self, '_data_format', 'NHWC') == 'NCHW'
        if nchw:                                       # images as preprocess_for_eval(..., data_format='NCHW') hands them over
            inputs = inputs.permute(0, 2, 3, 1)
        logits, objness_logits, localisations = self.forward_heads(inputs)
        fn = prediction_fn if prediction_fn is not None else ops.softmax_last
        predictions = [fn(l) for l in logits]
        objness_pred = [ops.softmax_last(o, pick=1) if prediction_fn is None else fn(o)[..., 1:2] for o in objness_logits]
        names = [n for n in (end_points or ()) if not (self.fuse_pools and n in ('block1', 'block2', 'block3'))]
        eps = {name: self.end_point(name, inputs.shape[0]) for name in names}
        if nchw:                                       # the feature maps in the caller's layout; the heads are channel-last either way (:401, :412)
            eps = {k: v.permute(0, 3, 1, 2).contiguous() for k, v in eps.items()}
        return predictions, logits, objness_pred, objness_logits, localisations, eps

    def end_point(self, name, n):
        shp = (C.c_int64 * 4)()
        rc = lib().ron_end_point_shape(self._context(), name.encode(), n, shp)
        if rc < 0:
            check(rc)
        out = torch.empty(tuple(shp), dtype=torch.float32, device=self.device)
        check(lib().ron_end_point_copy(self._context(), name.encode(), n, ptr(out), current_stream()))
        return out

    def arg_scope(self, weight_decay=0.0005, is_training=True, data_format='NHWC'):
        """Network arg_scope (nets/ron_vgg_320.py:156-159).  Layer defaults are baked into the HIP graph; the returned object is a
        context manager so `with slim.arg_scope(net.arg_scope(...))`-style code runs.  `data_format='NCHW'` (ron_eval.py:34 offers it):
        inside the context `net()` takes [N, 3, H, W] images and returns its end points as [N, C, H, W]; the kernels stay NHWC (the
        layout the MFMA tiles gather rows from), the image is re-laid on the way in."""
        return _DataFormatScope([self], data_format)

    def anchors(self, img_shape, dtype=np.float32):
        """Default anchor boxes (nets/ron_vgg_320.py:162-171): list of (y, x, h, w) per layer."""
        p = self.params
        return [ops.anchor_one_layer(img_shape, s, p.anchor_sizes[i], p.anchor_ratios[i], p.anchor_steps[i],
                                     offset=p.anchor_offset, dtype=dtype) for i, s in enumerate(p.feat_shapes)]

    def _anchors_device(self):
        if self._anchors_dev is None:
            self._anchors_dev = ops.anchors_to_device(self.anchors(self.params.img_shape), self.device)
        return self._anchors_dev

    def bboxes_decode(self, feat_localizations, anchors, scope='ssd_bboxes_decode'):
        """nets/ron_vgg_320.py:188-195 -> ssd_common.tf_ssd_bboxes_decode."""
        adev = ops.anchors_to_device(anchors, self.device)
        return [ops.bboxes_decode_layer(l, a, tuple(self.params.prior_scaling)) for l, a in zip(feat_localizations, adev)]

    def bboxes_filter_min(self, scores, bboxes, top_k, minsize=0.03, scope=None):
        """nets/ron_vgg_320.py:196-233: drop the boxes with w <= minsize or h <= minsize, keeping the order, zero-pad to top_k.
        Tensors ([1, N] / [1, N, 4]) or dicts class -> tensors, like the reference."""
        if isinstance(scores, dict) or isinstance(bboxes, dict):
            d_scores, d_bboxes = {}, {}
            for c in scores.keys():
                d_scores[c], d_bboxes[c] = self.bboxes_filter_min(scores[c], bboxes[c], top_k, minsize=minsize)
            return d_scores, d_bboxes
        return ops.bboxes_filter_min(scores, bboxes, top_k, minsize=minsize)

    def detected_bboxes(self, predictions, localisations, select_threshold=None, nms_threshold=0.5,
                        clipping_bbox=None, top_k=400, keep_top_k=200, nms_mode='min'):
        """nets/ron_vgg_320.py:234-256: per-class select -> clip -> filter_min -> sort -> NMS (TF semantics).
        Returns (dict_scores, dict_bboxes): class -> [N, keep_top_k] / [N, keep_top_k, 4], zero padded."""
        from .. import tfe
        return tfe.detected_bboxes(predictions, localisations, num_classes=self.params.num_classes,
                                   select_threshold=select_threshold, nms_threshold=nms_threshold,
                                   clipping_bbox=clipping_bbox, top_k=top_k, keep_top_k=keep_top_k,
                                   nms_mode=nms_mode, min_size=0.03)

    # ------------------------------------------------------------------ fused graded path
    def detect(self, inputs, objectness_thres=0.03, select_threshold=0.01, nms_threshold=0.45, top_k=400,
               bbox_img=(0., 0., 1., 1.), out=None):
        """forward + np_methods post-processing in one enqueue (ron_detect).  Returns DetectionBuffers
        (`out`, when given, is reused: n and capacity must match)."""
        if getattr(self, '_data_format', 'NHWC') == 'NCHW':
            inputs = inputs.permute(0, 2, 3, 1)
        inputs = inputs.to(self.device, torch.float32).contiguous()
        n = inputs.shape[0]
        cfg = _lib.PostCfg()
        cfg.objectness_thres, cfg.select_threshold, cfg.nms_threshold, cfg.top_k = \
            objectness_thres, (select_threshold or 0.0), nms_threshold, top_k
        for i in range(4):
            cfg.bbox_img[i] = bbox_img[i]
            cfg.prior_scaling[i] = self.params.prior_scaling[i]
        if out is None:
            out = ops.DetectionBuffers(n, top_k, self.device)
        assert out.n == n and out.capacity == top_k
        oc = out.c_struct()
        check(lib().ron_detect(self._context(), ptr(inputs), n, C.byref(cfg), C.byref(oc), current_stream()))
        return out


# ---------------------------------------------------------------------- the reference's function entries
# nets_factory.networks_map / arg_scopes_map point at module-level functions (nets/nets_factory.py:34-52): ron_net builds the graph
# under a variable scope, `reuse=True` finds the variables of an earlier call.  Here a scope name owns one network object (its packed
# weights are the scope's variables): the first call of a scope needs `weights=` (dict by TF variable name), later calls reuse it.
_SCOPES = {}


def _scoped_net(key, make, weights, reuse):
    net = _SCOPES.get(key)
    if net is None:
        if reuse:
            raise ValueError('Variable scope %s does not exist (reuse=True before the first call)' % (key[0],))
        if weights is None:
            raise ValueError('the first call of scope %r needs weights= (a dict keyed by TF variable names)' % (key[0],))
        net = _SCOPES[key] = make()
        net.load_weights(weights)
    elif weights is not None and not reuse:
        net.load_weights(weights)
    return net


def _ron_net_fn(variant, inputs, num_classes, feat_layers, anchor_sizes, anchor_ratios, is_training, dropout_keep_prob, prediction_fn,
                reuse, scope, weights, dtype, max_batch):
    d = RONNet.default_params
    params = d._replace(num_classes=num_classes, feat_layers=list(feat_layers), anchor_sizes=list(anchor_sizes), anchor_ratios=list(anchor_ratios))
    dev = inputs.device if inputs.is_cuda else torch.device('cuda', torch.cuda.current_device())
    key = (scope, variant, num_classes, dtype, str(dev))
    net = _scoped_net(key, lambda: RONNet(params, variant=variant, dtype=dtype, max_batch=max(max_batch, inputs.shape[0]), device=dev), weights, reuse)
    fmt = _ARG_SCOPE_FORMAT[-1] if _ARG_SCOPE_FORMAT else 'NHWC'
    with _DataFormatScope([net], fmt):
        return net.net(inputs, is_training=is_training, dropout_keep_prob=dropout_keep_prob, prediction_fn=prediction_fn, reuse=reuse, scope=scope)


def ron_net(inputs, num_classes=RONNet.default_params.num_classes, feat_layers=RONNet.default_params.feat_layers,
            anchor_sizes=RONNet.default_params.anchor_sizes, anchor_ratios=RONNet.default_params.anchor_ratios, is_training=True,
            dropout_keep_prob=0.5, prediction_fn=None, reuse=None, scope='ron_320_vgg', weights=None, dtype='bf16', max_batch=32):
    """RON net definition, full VGG-16 fc6 / fc7 (nets/ron_vgg_320.py:434-508): the reference's 6-tuple."""
    return _ron_net_fn('full', inputs, num_classes, feat_layers, anchor_sizes, anchor_ratios, is_training, dropout_keep_prob,
                       prediction_fn, reuse, scope, weights, dtype, max_batch)


def ron_net_reducedfc(inputs, num_classes=RONNet.default_params.num_classes, feat_layers=RONNet.default_params.feat_layers,
                      anchor_sizes=RONNet.default_params.anchor_sizes, anchor_ratios=RONNet.default_params.anchor_ratios, is_training=True,
                      dropout_keep_prob=0.5, prediction_fn=None, reuse=None, scope='ron_320_vgg', weights=None, dtype='bf16', max_batch=32):
    """RON net definition with the reduced fc6 / fc7 (nets/ron_vgg_320.py:510-580)."""
    return _ron_net_fn('reducedfc', inputs, num_classes, feat_layers, anchor_sizes, anchor_ratios, is_training, dropout_keep_prob,
                       prediction_fn, reuse, scope, weights, dtype, max_batch)


ron_net.default_image_size = 320

_ARG_SCOPE_FORMAT = []       # data formats of the arg scopes entered, innermost last


class _ArgScope(object):
    """ron_arg_scope / ssd_arg_scope: the layer defaults (SAME padding, ReLU, BN eps 1e-5, ...; nets/ron_vgg_320.py:595-629) are what
    the HIP graph implements and are not configurable; the scope carries the one thing a caller can choose, the data format."""

    def __init__(self, weight_decay=0.0005, is_training=True, data_format='NHWC'):
        if data_format not in ('NHWC', 'NCHW'):
            raise ValueError('data_format must be NHWC or NCHW, not %r' % (data_format,))
        self.weight_decay, self.is_training, self.data_format = weight_decay, is_training, data_format

    def __enter__(self):
        _ARG_SCOPE_FORMAT.append(self.data_format)
        return self

    def __exit__(self, *exc):
        _ARG_SCOPE_FORMAT.pop()
        return False


def ron_arg_scope(weight_decay=0.0005, is_training=True, data_format='NHWC'):
    """Defines the RON arg scope (nets/ron_vgg_320.py:595-629)."""
    return _ArgScope(weight_decay, is_training, data_format)
