"""SSD-512 (VGG) network object with the interface of the reference's ``nets/ssd_vgg_512.py`` (BASELINE config 5).

``SSDNet`` keeps names / defaults / return arity of the reference class (nets/ssd_vgg_512.py:63-218): ``net`` returns
``(predictions, localisations, logits, end_points)``; ``detected_bboxes`` is select -> top-k sort -> NMS with no clipping
and no min-size filter (:182-201).  Everything runs through libron_hip.so (ron_ctx variant RON_VARIANT_SSD512)."""
import contextlib
import ctypes as C
from collections import namedtuple

import numpy as np
import torch

from .. import _lib, ops
from .._lib import check, current_stream, lib, ptr
from . import ron_vgg_320
from .ron_vgg_320 import RONNet

# same field names as the reference namedtuple (nets/ssd_vgg_512.py:44-60)
SSDParams = namedtuple('SSDParameters', ['img_shape', 'num_classes', 'no_annotation_label', 'feat_layers', 'feat_shapes',
                                         'anchor_size_bounds', 'anchor_sizes', 'anchor_ratios', 'anchor_steps',
                                         'anchor_offset', 'normalizations', 'prior_scaling'])


class SSDNet(RONNet):
    """SSD VGG-based 512 network: conv4 64x64, conv7 32x32, conv8 16x16, conv9 8x8, conv10 4x4, conv11 2x2, conv12 1x1."""
    default_params = SSDParams(
        img_shape=(512, 512),
        num_classes=21,
        no_annotation_label=21,
        feat_layers=['block4', 'block7', 'block8', 'block9', 'block10', 'block11', 'block12'],
        feat_shapes=[(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)],
        anchor_size_bounds=[0.10, 0.90],
        anchor_sizes=[(20.48, 51.2), (51.2, 133.12), (133.12, 215.04), (215.04, 296.96), (296.96, 378.88),
                      (378.88, 460.8), (460.8, 542.72)],
        anchor_ratios=[[2, .5], [2, .5, 3, 1. / 3], [2, .5, 3, 1. / 3], [2, .5, 3, 1. / 3], [2, .5, 3, 1. / 3], [2, .5], [2, .5]],
        anchor_steps=[8, 16, 32, 64, 128, 256, 512],
        anchor_offset=0.5,
        normalizations=[20, -1, -1, -1, -1, -1, -1],
        prior_scaling=[0.1, 0.1, 0.2, 0.2])

    def __init__(self, params=None, dtype='bf16', max_batch=16, device=None, fuse_pools=False):
        self.params = params if isinstance(params, SSDParams) else SSDNet.default_params
        if dtype not in _lib.DTYPES:
            raise ValueError('Unknown dtype %s' % dtype)
        self.variant, self.dtype, self.max_batch, self.fuse_pools = 'ssd512', dtype, max_batch, fuse_pools
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self._ctx = None
        self._anchors_dev = None

    def _head_buffers(self, n):
        geom, nc = self._head_geometry()             # from the context, not from params.feat_shapes (update_feature_shapes)
        cls, loc = [], []
        for (fh, fw, a) in geom:
            cls.append(torch.empty((n, fh, fw, a, nc), dtype=torch.float32, device=self.device))
            loc.append(torch.empty((n, fh, fw, a, 4), dtype=torch.float32, device=self.device))
        return cls, None, loc

    def forward_heads(self, inputs):
        """Conv stack only: (logits, None, localisations)."""
        inputs = inputs.to(self.device, torch.float32).contiguous()
        n = inputs.shape[0]
        assert tuple(inputs.shape[1:]) == tuple(self.params.img_shape) + (3,)
        cls, _, loc = self._head_buffers(n)
        hd = _lib.Heads()
        for i in range(len(cls)):
            hd.cls[i], hd.loc[i] = cls[i].data_ptr(), loc[i].data_ptr()
        check(lib().ron_forward(self._context(), ptr(inputs), n, C.byref(hd), current_stream()))
        return cls, None, loc

    def net(self, inputs, is_training=True, update_feat_shapes=True, dropout_keep_prob=0.5, prediction_fn=None, reuse=None,
            scope='ssd_512_vgg', end_points=('block4', 'block7', 'block8', 'block9', 'block10', 'block11', 'block12')):
        """nets/ssd_vgg_512.py:108-133 -> (predictions, localisations, logits, end_points)."""
        nchw = getattr(self, '_data_format', 'NHWC') == 'NCHW'
        if nchw:
            inputs = inputs.permute(0, 2, 3, 1)
        logits, _, localisations = self.forward_heads(inputs)
        fn = prediction_fn if prediction_fn is not None else ops.softmax_last
        predictions = [fn(l) for l in logits]
        eps = {name: self.end_point(name, inputs.shape[0]) for name in (end_points or ())}
        if nchw:
            eps = {k: v.permute(0, 3, 1, 2).contiguous() for k, v in eps.items()}
        if update_feat_shapes:                       # nets/ssd_vgg_512.py:134-136: the feature shapes follow the predictions
            self.update_feature_shapes(predictions)
        return predictions, localisations, logits, eps

    def update_feature_shapes(self, predictions):
        """nets/ssd_vgg_512.py:141-146 -> ssd_feat_shapes_from_net (nets/ssd_vgg_300.py:282-303): feat_shapes become the
        [H, W, A] of every prediction tensor ([N, H, W, A, C]); anchors() reads the first two entries like the reference's."""
        self.params = self.params._replace(feat_shapes=[[int(d) for d in p.shape[1:4]] for p in predictions])

    def anchors(self, img_shape, dtype=np.float32):
        """nets/ssd_vgg_512.py:148-157: list of (y, x, h, w) per layer."""
        p = self.params
        out = []
        for i, s in enumerate(p.feat_shapes):
            fh, fw = int(s[0]), int(s[1])
            sizes, ratios = p.anchor_sizes[i], p.anchor_ratios[i]
            na = len(sizes) + len(ratios)
            y = np.empty((fh, fw, 1), np.float32)
            x = np.empty((fh, fw, 1), np.float32)
            h = np.empty((na,), np.float32)
            w = np.empty((na,), np.float32)
            sz = (C.c_double * len(sizes))(*[float(v) for v in sizes])
            rt = (C.c_double * max(len(ratios), 1))(*[float(v) for v in ratios])
            check(lib().ron_ssd_anchor_one_layer(int(img_shape[0]), int(img_shape[1]), fh, fw, sz, len(sizes), rt, len(ratios),
                                                 float(p.anchor_steps[i]), float(p.anchor_offset), ptr(y), ptr(x), ptr(h), ptr(w)))
            out.append((y.astype(dtype, copy=False), x.astype(dtype, copy=False), h.astype(dtype, copy=False), w.astype(dtype, copy=False)))
        return out

    def detected_bboxes(self, predictions, localisations, select_threshold=None, nms_threshold=0.5, clipping_bbox=None,
                        top_k=400, keep_top_k=200, nms_mode='min'):
        """nets/ssd_vgg_512.py:182-201: select -> sort -> NMS; `clipping_bbox` is accepted and ignored like there."""
        from .. import tfe
        return tfe.detected_bboxes(predictions, localisations, num_classes=self.params.num_classes,
                                   select_threshold=select_threshold, nms_threshold=nms_threshold, clipping_bbox=None,
                                   top_k=top_k, keep_top_k=keep_top_k, nms_mode=nms_mode, min_size=None)

    def detect(self, inputs, select_threshold=0.01, nms_threshold=0.45, top_k=400, bbox_img=(0., 0., 1., 1.), out=None):
        """forward + np_methods post-processing (no objectness gate for SSD)."""
        return RONNet.detect(self, inputs, objectness_thres=0.0, select_threshold=select_threshold,
                             nms_threshold=nms_threshold, top_k=top_k, bbox_img=bbox_img, out=out)


# ---------------------------------------------------------------------- the reference's function entries (nets_factory.networks_map)
def ssd_net(inputs, num_classes=SSDNet.default_params.num_classes, feat_layers=SSDNet.default_params.feat_layers,
            anchor_sizes=SSDNet.default_params.anchor_sizes, anchor_ratios=SSDNet.default_params.anchor_ratios,
            normalizations=SSDNet.default_params.normalizations, is_training=True, dropout_keep_prob=0.5, prediction_fn=None,
            reuse=None, scope='ssd_512_vgg', weights=None, dtype='bf16', max_batch=16):
    """SSD net definition (nets/ssd_vgg_512.py:364-460): (predictions, localisations, logits, end_points).  A scope name owns one
    network object, like ron_vgg_320.ron_net: the first call needs `weights=`."""
    params = SSDNet.default_params._replace(num_classes=num_classes, feat_layers=list(feat_layers), anchor_sizes=list(anchor_sizes),
                                            anchor_ratios=list(anchor_ratios), normalizations=list(normalizations))
    dev = inputs.device if inputs.is_cuda else torch.device('cuda', torch.cuda.current_device())
    key = (scope, 'ssd512', num_classes, dtype, str(dev))
    net = ron_vgg_320._scoped_net(key, lambda: SSDNet(params, dtype=dtype, max_batch=max(max_batch, inputs.shape[0]), device=dev), weights, reuse)
    fmt = ron_vgg_320._ARG_SCOPE_FORMAT[-1] if ron_vgg_320._ARG_SCOPE_FORMAT else 'NHWC'
    with ron_vgg_320._DataFormatScope([net], fmt):
        return net.net(inputs, is_training=is_training, update_feat_shapes=False, dropout_keep_prob=dropout_keep_prob,
                       prediction_fn=prediction_fn, reuse=reuse, scope=scope)


ssd_net.default_image_size = 512


def ssd_arg_scope(weight_decay=0.0005, data_format='NHWC'):
    """Defines the VGG arg scope (nets/ssd_vgg_512.py:463-487)."""
    return ron_vgg_320._ArgScope(weight_decay, True, data_format)


def ssd_arg_scope_caffe(caffe_scope):
    """nets/ssd_vgg_512.py:490-513 takes the Caffe weight initialisers from `caffe_scope`; initialisers have no meaning for an
    inference graph whose weights are loaded: the scope is the plain one."""
    return ron_vgg_320._ArgScope()
