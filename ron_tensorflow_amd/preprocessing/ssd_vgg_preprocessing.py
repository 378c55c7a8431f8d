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


def preprocess_for_eval_batch(images, out_shape=EVAL_SIZE, resize=Resize.WARP_RESIZE, device='cuda:0',
                              means=(_R_MEAN, _G_MEAN, _B_MEAN)):
    """List of HWC uint8 images (numpy or torch, any sizes) -> float32 GPU tensor [N, out_h, out_w, 3]: one packed
    upload, one launch."""
    if resize not in (Resize.WARP_RESIZE, Resize.NONE):
        raise NotImplementedError('only Resize.WARP_RESIZE / Resize.NONE (what the eval drivers use) run on the GPU')
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
    with torch.cuda.device(dev):
        check(lib().ron_preprocess_eval(ptr(packed), ptr(d_off), ptr(d_hw), len(arrs), int(out_shape[0]), int(out_shape[1]), m,
                                        ptr(out), current_stream()))
    return out


def preprocess_for_eval(image, labels, bboxes, out_shape=EVAL_SIZE, data_format='NHWC', difficults=None,
                        resize=Resize.WARP_RESIZE, device='cuda:0'):
    """Reference signature for one image: returns (image [out_h, out_w, 3] float32 GPU, labels, bboxes, bbox_img).
    Difficult ground truth is removed when ``difficults`` is given (:415-419); bboxes are unchanged by a warp."""
    img = preprocess_for_eval_batch([image], out_shape, resize, device)[0]
    if data_format == 'NCHW':
        img = img.permute(2, 0, 1).contiguous()
    bbox_img = np.array([0., 0., 1., 1.], np.float32)
    if difficults is not None and labels is not None:
        mask = ~np.asarray(difficults).astype(bool)
        labels = np.asarray(labels)[mask]
        bboxes = np.asarray(bboxes)[mask]
    return img, labels, bboxes, bbox_img
