"""CPU: identities anchoring the restated TF1 bilinear resize (oracle/preprocess.py, parity unpinned by the reference)."""
import numpy as np

from oracle import preprocess as op


def test_same_size_is_whitening_only():
    rs = np.random.RandomState(0)
    im = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    out = op.preprocess_for_eval(im, (37, 53))
    assert np.array_equal(out, im.astype(np.float32) - np.array(op.MEANS, np.float32))


def test_integer_downscale_samples_pixels():
    rs = np.random.RandomState(1)
    im = rs.randint(0, 256, (640, 960, 3)).astype(np.uint8)
    out = op.preprocess_for_eval(im, (320, 320))                 # scale 2 x 3: legacy coordinates land on pixels
    assert np.array_equal(out, im[::2, ::3].astype(np.float32) - np.array(op.MEANS, np.float32))


def test_constant_image_and_upscale_bounds():
    im = np.full((50, 70, 3), 200, np.uint8)
    out = op.preprocess_for_eval(im, (320, 320))
    assert np.array_equal(out, np.broadcast_to(np.float32(200) - np.array(op.MEANS, np.float32), out.shape))
    rs = np.random.RandomState(2)
    im = rs.randint(0, 256, (100, 120, 3)).astype(np.uint8)
    out = op.preprocess_for_eval(im, (320, 320))
    w = im.astype(np.float32) - np.array(op.MEANS, np.float32)
    assert out.min() >= w.min() - 1e-3 and out.max() <= w.max() + 1e-3
    assert np.array_equal(out[0, 0], w[0, 0])
    # x2 upscale of a ramp: legacy (no half-pixel) coordinates give exact midpoints
    ramp = np.tile(np.arange(0, 160, dtype=np.uint8)[None, :, None], (160, 1, 3))
    out = op.preprocess_for_eval(ramp, (320, 320), means=(0, 0, 0))
    assert np.array_equal(out[5, :319, 0], np.arange(319, dtype=np.float32) * np.float32(0.5))
    assert out[5, 319, 0] == 159.0                                # clamped upper neighbour


def test_crop_or_pad_and_modes():
    rs = np.random.RandomState(3)
    im = rs.randint(0, 256, (400, 200, 3)).astype(np.uint8)        # taller (cropped) and narrower (padded) than 320 x 320
    w = im.astype(np.float32) - np.array(op.MEANS, np.float32)
    out, b, rect = op.preprocess_for_eval_mode(im, [[0., 0., 1., 1.], [.25, .5, .75, 1.]], (320, 320), 'CENTRAL_CROP')
    assert np.array_equal(out[:, 60:260], w[40:360]) and not out[:, :60].any() and not out[:, 260:].any()
    assert np.allclose(rect, [-40 / 320., 60 / 320., 360 / 320., 260 / 320.])
    assert np.allclose(b[1], [(100 - 40) / 320., (100 + 60) / 320., (300 - 40) / 320., (200 + 60) / 320.])
    out, b, rect = op.preprocess_for_eval_mode(im, [], (320, 320), 'PAD_AND_RESIZE')     # factor 0.8 -> 320 x 160, padded to 320 wide
    assert np.array_equal(out[:, 80:240], op.resize_bilinear(w, (320, 160))) and not out[:, :80].any()
    assert np.allclose(rect, [0., .25, 1., .75])
    small = rs.randint(0, 256, (100, 120, 3)).astype(np.uint8)     # smaller than the target: factor 1, only padded
    out, _, _ = op.preprocess_for_eval_mode(small, [], (320, 320), 'PAD_AND_RESIZE')
    assert np.array_equal(out[110:210, 100:220], small.astype(np.float32) - np.array(op.MEANS, np.float32))
    out, _, rect = op.preprocess_for_eval_mode(small, [], (320, 320), 'NONE')
    assert out.shape == (100, 120, 3) and rect.tolist() == [0., 0., 1., 1.]


def test_resize_matches_an_independent_bilinear_sampler():
    """The restated TF1 resize against torch's grid_sample (an independent bilinear engine) fed with TF1's sampling positions: source
    coordinate = out_index * (in_size / out_size), no half-pixel offset, indices clamped at the last pixel (align_corners=False of
    TensorFlow 1.x, tf_image.py:269-282 -> tf.image.resize_images).  Pins the interpolation arithmetic of oracle/preprocess.py; the
    coordinate convention itself stays what the module header says it is: TensorFlow's published one, restated."""
    import torch
    import torch.nn.functional as F
    from oracle import preprocess as op
    rs = np.random.RandomState(3)
    for (h, w), (oh, ow) in (((375, 500), (320, 320)), ((333, 500), (320, 320)), ((97, 61), (320, 320)), ((512, 640), (512, 512)), ((40, 40), (7, 9))):
        img = rs.uniform(-130, 130, (h, w, 3)).astype(np.float32)
        ref = op.resize_bilinear(img, (oh, ow))
        sy = np.arange(oh, dtype=np.float32) * (np.float32(h) / np.float32(oh))
        sx = np.arange(ow, dtype=np.float32) * (np.float32(w) / np.float32(ow))
        gy = 2.0 * sy.astype(np.float64) / max(h - 1, 1) - 1.0           # grid_sample, align_corners=True: -1 -> pixel 0, +1 -> pixel size - 1
        gx = 2.0 * sx.astype(np.float64) / max(w - 1, 1) - 1.0
        grid = np.stack(np.meshgrid(gx, gy), -1)[None]                     # [1, oh, ow, (x, y)]
        t = torch.from_numpy(img.transpose(2, 0, 1)[None].astype(np.float64))
        got = F.grid_sample(t, torch.from_numpy(grid), mode='bilinear', padding_mode='border', align_corners=True)[0].numpy().transpose(1, 2, 0)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-4, ((h, w), (oh, ow), np.abs(got - ref).max())
