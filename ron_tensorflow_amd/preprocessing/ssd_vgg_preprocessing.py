"""Evaluation preprocessing on the GPU, reference names:
``preprocess_for_eval`` (preprocessing/ssd_vgg_preprocessing.py:358-425) for the eval driver's resize mode
(``Resize.WARP_RESIZE``, eval_ron_network.py:150-158) and ``Resize.NONE``.  JPEG decode stays on the host: the
input here is the decoded uint8 RGB image."""
import ctypes as C

import numpy as np
import torch

from .._lib import check, current_stream, lib, ptr

_R_MEAN, _G_MEAN, _B_MEAN = 123., 117., 104.          # ssd_vgg_preprocessing.py:30-32
EVAL_SIZE = (320, 320)


class Resize(object):                                 # ssd_vgg_preprocessing.py:22-27 (IntEnum there)
    NONE, CENTRAL_CROP, PAD_AND_RESIZE, WARP_RESIZE = 0, 1, 2, 3


def eval_geometry(h, w, out_shape, resize):
    """Geometry of one image for a resize mode: ({crop_y, crop_x, crop_h, crop_w, pad_y, pad_x, resized_h, resized_w},
    bbox transform).  The bbox transform maps relative (ymin, xmin, ymax, xmax) of the source image to the output the
    way tf_image.bboxes_crop_or_pad does (tf_image.py:141-166): returns (scale[4], offset[4]) with b' = b * scale + offset."""
    oh, ow = int(out_shape[0]), int(out_shape[1])
    one, zero = np.ones(4, np.float64), np.zeros(4, np.float64)
    if resize == Resize.WARP_RESIZE:
        return (0, 0, h, w, 0, 0, oh, ow), (one, zero)
    if resize == Resize.NONE:
        return (0, 0, h, w, 0, 0, h, w), (one, zero)
    if resize == Resize.PAD_AND_RESIZE:                                   # ssd_vgg_preprocessing.py:392-405
        factor = min(1.0, min(oh / h, ow / w))
        rh, rw = int(np.floor(factor * h)), int(np.floor(factor * w))
    elif resize == Resize.CENTRAL_CROP:
        rh, rw = h, w
    else:
        raise ValueError('unknown resize mode %r' % (resize,))
    # tf_image.resize_image_bboxes_with_crop_or_pad on the rh x rw image (tf_image.py:169-254)
    wd, hd = ow - rw, oh - rh
    crop_x, pad_x = max(-wd // 2, 0), max(wd // 2, 0)
    crop_y, pad_y = max(-hd // 2, 0), max(hd // 2, 0)
    hc, wc = min(oh, rh), min(ow, rw)
    # the crop is taken in resized coordinates; the kernel resizes the crop window of the SOURCE: scale 1 for CENTRAL_CROP, and
    # PAD_AND_RESIZE never crops (rh <= oh, rw <= ow)
    if resize == Resize.PAD_AND_RESIZE:
        geom = (0, 0, h, w, pad_y, pad_x, rh, rw)
    else:
        geom = (crop_y, crop_x, hc, wc, pad_y, pad_x, hc, wc)
    s1 = np.array([rh, rw, rh, rw], np.float64)
    o1 = np.array([-crop_y, -crop_x, -crop_y, -crop_x], np.float64)
    s2 = np.array([hc, wc, hc, wc], np.float64)
    o2 = np.array([pad_y, pad_x, pad_y, pad_x], np.float64)
    t = np.array([oh, ow, oh, ow], np.float64)
    # b -> ((b * s1 + o1) / s2 * s2 + o2) / t
    return geom, (s1 / t, (o1 + o2) / t)


def preprocess_for_eval_batch(images, out_shape=EVAL_SIZE, resize=Resize.WARP_RESIZE, device='cuda:0',
                              means=(_R_MEAN, _G_MEAN, _B_MEAN)):
    """List of HWC uint8 images (numpy or torch, any sizes) -> float32 GPU tensor [N, out_h, out_w, 3]: one packed
    upload, one launch.  All four modes of the reference (Resize.NONE needs equally sized images)."""
    if resize not in (Resize.WARP_RESIZE, Resize.NONE, Resize.CENTRAL_CROP, Resize.PAD_AND_RESIZE):
        raise ValueError('unknown resize mode %r' % (resize,))
    dev = torch.device(device)
    arrs = []
    for im in images:
        a = im.detach().cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError('Input must be of size [height, width, C>0]')        # :374 (C = 3 on this path)
        if a.dtype != np.uint8:
            raise ValueError('decoded images are uint8')
        arrs.append(np.ascontiguousarray(a))
    hw = np.array([[a.shape[0], a.shape[1]] for a in arrs], np.int32)
    if resize == Resize.NONE:
        if len(set(map(tuple, hw.tolist()))) != 1:
            raise ValueError('Resize.NONE needs equally sized images in a batch')
        out_shape = (int(hw[0, 0]), int(hw[0, 1]))
    sizes = np.array([a.size for a in arrs], np.int64)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    packed = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    d_hw = torch.from_numpy(hw).to(dev)
    out = torch.empty((len(arrs), int(out_shape[0]), int(out_shape[1]), 3), dtype=torch.float32, device=dev)
    m = (C.c_float * 3)(*means)
    d_geom = None
    if resize in (Resize.CENTRAL_CROP, Resize.PAD_AND_RESIZE):
        geom = np.array([eval_geometry(a.shape[0], a.shape[1], out_shape, resize)[0] for a in arrs], np.int32)
        d_geom = torch.from_numpy(geom).to(dev)
    with torch.cuda.device(dev):
        check(lib().ron_preprocess_eval_geom(ptr(packed), ptr(d_off), ptr(d_hw), ptr(d_geom), len(arrs), int(out_shape[0]),
                                             int(out_shape[1]), m, ptr(out), current_stream()))
    return out


def preprocess_for_eval(image, labels, bboxes, out_shape=EVAL_SIZE, data_format='NHWC', difficults=None,
                        resize=Resize.WARP_RESIZE, device='cuda:0'):
    """Reference signature for one image: returns (image [out_h, out_w, 3] float32 GPU, labels, bboxes, bbox_img).
    Difficult ground truth is removed when ``difficults`` is given (:415-419); bboxes are unchanged by a warp and follow
    the crop / pad otherwise, as does ``bbox_img`` (the image rectangle, :379-384, :413-414)."""
    img = preprocess_for_eval_batch([image], out_shape, resize, device)[0]
    if data_format == 'NCHW':
        img = img.permute(2, 0, 1).contiguous()
    a = image.detach().cpu().numpy() if isinstance(image, torch.Tensor) else np.asarray(image)
    _, (scale, offset) = eval_geometry(a.shape[0], a.shape[1], a.shape[:2] if resize == Resize.NONE else out_shape, resize)
    bbox_img = (np.array([0., 0., 1., 1.]) * scale + offset).astype(np.float32)
    if bboxes is not None:
        bboxes = (np.asarray(bboxes, np.float64).reshape(-1, 4) * scale + offset).astype(np.float32)
    if difficults is not None and labels is not None:
        mask = ~np.asarray(difficults).astype(bool)
        labels = np.asarray(labels)[mask]
        bboxes = np.asarray(bboxes)[mask]
    return img, labels, bboxes, bbox_img
