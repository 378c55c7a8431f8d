"""GPU: ron_preprocess_eval vs the oracle restatement of preprocess_for_eval (WARP_RESIZE) -- bit-exact float32."""
import numpy as np
import pytest
import torch

from oracle import preprocess as op

pytestmark = pytest.mark.gpu


def test_ragged_batch_bit_exact():
    from ron_tensorflow_amd.preprocessing import ssd_vgg_preprocessing as pp
    rs = np.random.RandomState(5)
    shapes = [(375, 500), (500, 333), (320, 320), (1, 1), (97, 1024), (640, 960), (2, 3)]
    ims = [rs.randint(0, 256, s + (3,)).astype(np.uint8) for s in shapes]
    out = pp.preprocess_for_eval_batch(ims).cpu().numpy()
    assert out.shape == (len(ims), 320, 320, 3)
    for i, im in enumerate(ims):
        assert np.array_equal(out[i], op.preprocess_for_eval(im, (320, 320))), 'image %d %r' % (i, shapes[i])


def test_other_out_shape_and_none():
    from ron_tensorflow_amd.preprocessing import ssd_vgg_preprocessing as pp
    rs = np.random.RandomState(6)
    ims = [rs.randint(0, 256, (300, 400, 3)).astype(np.uint8) for _ in range(3)]
    out = pp.preprocess_for_eval_batch(ims, out_shape=(512, 512)).cpu().numpy()
    for i, im in enumerate(ims):
        assert np.array_equal(out[i], op.preprocess_for_eval(im, (512, 512)))
    out = pp.preprocess_for_eval_batch(ims, resize=pp.Resize.NONE).cpu().numpy()
    for i, im in enumerate(ims):
        assert np.array_equal(out[i], im.astype(np.float32) - np.array([123., 117., 104.], np.float32))
    with pytest.raises(ValueError):
        pp.preprocess_for_eval_batch(ims, resize=17)
    with pytest.raises(ValueError):
        pp.preprocess_for_eval_batch([np.zeros((4, 4), np.uint8)])


def test_reference_signature_and_feeds_the_net():
    from ron_tensorflow_amd.nets import nets_factory
    from ron_tensorflow_amd.preprocessing import ssd_vgg_preprocessing as pp
    from ron_tensorflow_amd import weights
    rs = np.random.RandomState(7)
    im = rs.randint(0, 256, (375, 500, 3)).astype(np.uint8)
    labels = np.array([3, 7, 9]); bboxes = rs.rand(3, 4).astype(np.float32); diff = np.array([0, 1, 0])
    img, l2, b2, bbox_img = pp.preprocess_for_eval(im, labels, bboxes, difficults=diff)
    assert tuple(img.shape) == (320, 320, 3) and l2.tolist() == [3, 9] and b2.shape == (2, 4)
    assert bbox_img.tolist() == [0., 0., 1., 1.]
    nchw, _, _, _ = pp.preprocess_for_eval(im, None, None, data_format='NCHW')
    assert tuple(nchw.shape) == (3, 320, 320) and torch.equal(nchw.permute(1, 2, 0), img)
    net = nets_factory.get_network('ron_320_vgg')(variant='reducedfc', dtype='bf16', max_batch=1)
    net.load_weights(weights.synthetic_weights('reducedfc', seed=1))
    dets = net.detect(img[None])
    assert dets.count.shape[0] == 1
    net.close()


@pytest.mark.parametrize('mode', ['CENTRAL_CROP', 'PAD_AND_RESIZE'])
def test_crop_and_pad_modes_bit_exact(mode):
    """The two other modes of preprocess_for_eval: images larger, smaller and mixed relative to 320 x 320, with bboxes."""
    from ron_tensorflow_amd.preprocessing import ssd_vgg_preprocessing as pp
    rs = np.random.RandomState(8)
    shapes = [(375, 500), (200, 250), (500, 200), (320, 320), (321, 319), (97, 1024), (640, 960)]
    ims = [rs.randint(0, 256, s + (3,)).astype(np.uint8) for s in shapes]
    out = pp.preprocess_for_eval_batch(ims, resize=getattr(pp.Resize, mode)).cpu().numpy()
    bb = rs.rand(3, 4).astype(np.float32)
    for i, im in enumerate(ims):
        ref, ref_b, ref_rect = op.preprocess_for_eval_mode(im, bb, (320, 320), mode)
        assert np.array_equal(out[i], ref), 'image %d %r' % (i, shapes[i])
        img, _, b2, rect = pp.preprocess_for_eval(im, None, bb, resize=getattr(pp.Resize, mode))
        assert np.array_equal(img.cpu().numpy(), ref)
        assert np.abs(b2 - ref_b).max() <= 1e-6 and np.abs(rect - ref_rect).max() <= 1e-6
