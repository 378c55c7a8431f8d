"""Oracle: evaluation preprocessing (test infrastructure, see ``oracle/__init__.py``).

``preprocess_for_eval`` with WARP_RESIZE (preprocessing/ssd_vgg_preprocessing.py:358-425): to_float, subtract the
channel means (:41-55), then ``tf.image.resize_images(BILINEAR, align_corners=False)`` (tf_image.py:269-282).
The resize is TensorFlow 1.x's kernel, which is not in /root/reference (TF is a dependency, r1.x): its published
algorithm is restated here -- source coordinate ``out_index * (in_size / out_size)`` in float32, lower = floor,
upper = min(lower + 1, size - 1), ``top = tl + (tr - tl) * xl; bottom = bl + (br - bl) * xl; top + (bottom - top) * yl``.
**Parity unpinned** (no TF here to generate vectors); anchored on identities: size-preserving resize is exact,
integer down-scales sample source pixels exactly, constant images stay constant; the interpolation arithmetic is cross-checked against
an independent bilinear engine (torch grid_sample fed with these sampling positions, tests/test_oracle_preprocess.py).
"""
import numpy as np

F32 = np.float32
MEANS = (123., 117., 104.)


def preprocess_for_eval(image, out_shape=(320, 320), means=MEANS):
    img = np.asarray(image).astype(F32) - np.asarray(means, F32)
    h, w = img.shape[:2]
    oh, ow = out_shape
    sy, sx = F32(h) / F32(oh), F32(w) / F32(ow)
    in_y = np.arange(oh, dtype=F32) * sy
    in_x = np.arange(ow, dtype=F32) * sx
    y0 = np.floor(in_y).astype(np.int64); x0 = np.floor(in_x).astype(np.int64)
    y1 = np.minimum(y0 + 1, h - 1); x1 = np.minimum(x0 + 1, w - 1)
    ly = (in_y - y0.astype(F32))[:, None, None]
    lx = (in_x - x0.astype(F32))[None, :, None]
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    return (top + (bot - top) * ly).astype(F32)


def resize_bilinear(img, out_shape):
    """TF1 bilinear, align_corners=False, on a float32 HWC image (the core of preprocess_for_eval above)."""
    img = np.asarray(img, F32)
    h, w = img.shape[:2]
    oh, ow = out_shape
    sy, sx = F32(h) / F32(oh), F32(w) / F32(ow)
    in_y = np.arange(oh, dtype=F32) * sy
    in_x = np.arange(ow, dtype=F32) * sx
    y0 = np.floor(in_y).astype(np.int64); x0 = np.floor(in_x).astype(np.int64)
    y1 = np.minimum(y0 + 1, h - 1); x1 = np.minimum(x0 + 1, w - 1)
    ly = (in_y - y0.astype(F32))[:, None, None]
    lx = (in_x - x0.astype(F32))[None, :, None]
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * lx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * lx
    return (top + (bot - top) * ly).astype(F32)


def crop_or_pad(img, bboxes, th, tw):
    """tf_image.resize_image_bboxes_with_crop_or_pad (preprocessing/tf_image.py:169-254) on a float32 image + relative bboxes
    (tf_image.bboxes_crop_or_pad, :141-166)."""
    h, w = img.shape[:2]
    wd, hd = tw - w, th - h
    ocw, opw = max(-wd // 2, 0), max(wd // 2, 0)
    och, oph = max(-hd // 2, 0), max(hd // 2, 0)
    hc, wc = min(th, h), min(tw, w)
    out = np.zeros((th, tw, img.shape[2]), F32)
    out[oph:oph + hc, opw:opw + wc] = img[och:och + hc, ocw:ocw + wc]
    b = np.asarray(bboxes, np.float64).reshape(-1, 4)
    b = (b * np.array([h, w, h, w]) + np.array([-och, -ocw, -och, -ocw])) / np.array([hc, wc, hc, wc])
    b = (b * np.array([hc, wc, hc, wc]) + np.array([oph, opw, oph, opw])) / np.array([th, tw, th, tw])
    return out, b.astype(F32)


def preprocess_for_eval_mode(image, bboxes, out_shape=(320, 320), mode='WARP_RESIZE', means=MEANS):
    """preprocess_for_eval (ssd_vgg_preprocessing.py:358-425) for every resize mode -> (image, bboxes, bbox_img); the
    image rectangle [0,0,1,1] travels with the bboxes as their first row (:379-384, :413-414)."""
    img = np.asarray(image).astype(F32) - np.asarray(means, F32)
    b = np.concatenate([[[0., 0., 1., 1.]], np.asarray(bboxes, np.float64).reshape(-1, 4)], 0)
    if mode == 'NONE':
        pass
    elif mode == 'CENTRAL_CROP':
        img, b = crop_or_pad(img, b, out_shape[0], out_shape[1])
    elif mode == 'PAD_AND_RESIZE':
        h, w = img.shape[:2]
        factor = min(1.0, min(out_shape[0] / h, out_shape[1] / w))
        img = resize_bilinear(img, (int(np.floor(factor * h)), int(np.floor(factor * w))))
        img, b = crop_or_pad(img, b, out_shape[0], out_shape[1])
    elif mode == 'WARP_RESIZE':
        img = resize_bilinear(img, out_shape)
    else:
        raise ValueError(mode)
    b = np.asarray(b, F32)
    return img, b[1:], b[0]
