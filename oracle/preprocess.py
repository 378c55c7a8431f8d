"""Oracle: evaluation preprocessing (test infrastructure, see ``oracle/__init__.py``).

``preprocess_for_eval`` with WARP_RESIZE (preprocessing/ssd_vgg_preprocessing.py:358-425): to_float, subtract the
channel means (:41-55), then ``tf.image.resize_images(BILINEAR, align_corners=False)`` (tf_image.py:269-282).
The resize is TensorFlow 1.x's kernel, which is not in /root/reference (TF is a dependency, r1.x): its published
algorithm is restated here -- source coordinate ``out_index * (in_size / out_size)`` in float32, lower = floor,
upper = min(lower + 1, size - 1), ``top = tl + (tr - tl) * xl; bottom = bl + (br - bl) * xl; top + (bottom - top) * yl``.
**Parity unpinned** (no TF here to generate vectors); anchored on identities: size-preserving resize is exact,
integer down-scales sample source pixels exactly, constant images stay constant.
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
