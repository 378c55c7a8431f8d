"""GPU parity at class counts other than VOC's 21.

`RONParams.num_classes` is free in the reference (`default_params._replace(num_classes=N)` is the first call of
eval_ron_network.py:149, nets/np_methods.py:91-95 never looks at the count); the entry points take 2 ... RON_MAX_CLASSES = 128.
Checked at 2 (one foreground class: the whole list is one class-wise NMS segment), 21 and 81 (COCO's count: beyond the 64 the
class-wise scan was written for until round 5; the select kernels' staging tile then needs the raised dynamic-LDS limit):
ron_post_np bit-exact against the oracle, ron_post_tfe against oracle/tfe_post.py, the fp32 forward of `reducedfc` + detect."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import anchors as oanchors  # noqa: E402
from oracle import np_post, synth, tfe_post  # noqa: E402
from oracle import ron_forward as orf  # noqa: E402

CLASS_COUNTS = [2, 21, 81]


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _to_dev(lst, dev):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in lst]


def _bg(num_classes):
    # background offset that leaves a few thousand candidates above 0.01 whatever the count
    return {2: 3.0, 21: 7.0, 81: 6.0}[num_classes]


@pytest.mark.parametrize('num_classes', CLASS_COUNTS)
def test_post_np_bit_exact_at_other_class_counts(dev, num_classes):
    from ron_tensorflow_amd import ops
    anchors = oanchors.anchors_all_layers()
    adev = ops.anchors_to_device(anchors, dev)
    batch = 2
    cls, obj, loc = synth.head_tensors(300 + num_classes, batch=batch, num_classes=num_classes, bg=_bg(num_classes), ob=-2.0)
    loc = [l * np.float32(0.3) for l in loc]                                 # overlapping boxes: the NMS has work to do
    cls_d, obj_d, loc_d = _to_dev(cls, dev), _to_dev(obj, dev), _to_dev(loc, dev)
    pred_d = [ops.softmax_last(c) for c in cls_d]
    objp_d = [ops.softmax_last(o, pick=1) for o in obj_d]
    ref = np_post.detect_from_predictions([p.cpu().numpy() for p in pred_d], loc, anchors,
                                          objness_pred=[o.cpu().numpy() for o in objp_d], objectness_thres=0.03,
                                          select_threshold=0.01, top_k=400, nms_threshold=0.45)
    out, srt, ncand = ops.post_np(pred_d, objp_d, loc_d, adev, num_classes=num_classes, select_threshold=0.01, nms_threshold=0.45,
                                  cls_is_prob=True, obj_is_prob=True, want_sorted=True)
    got = out.to_lists()
    ncand = ncand.cpu().numpy()
    seen = set()
    for b in range(batch):
        assert ncand[b] == ref[b]['n_candidates'] and ncand[b] > 400                # the top-k cut is exercised
        assert np.array_equal(got[b]['classes'], ref[b]['classes'])
        assert np.array_equal(got[b]['anchor_index'], ref[b]['anchor_index'])
        assert np.array_equal(got[b]['scores'], ref[b]['scores'])
        np.testing.assert_allclose(got[b]['bboxes'], ref[b]['bboxes'], rtol=0, atol=1e-5)
        assert len(ref[b]['classes']) > 0 and len(ref[b]['classes']) < ref[b]['n_sorted']      # something kept, something suppressed
        seen |= set(int(c) for c in ref[b]['classes'])
    assert min(seen) >= 1 and max(seen) <= num_classes - 1
    if num_classes == 81:
        assert max(seen) >= 64                                              # class ids beyond the old limit take part
    # logits in, softmax + gate fused into the select kernel: the same lists
    out2, _, ncand2 = ops.post_np(cls_d, obj_d, loc_d, adev, num_classes=num_classes, select_threshold=0.01, nms_threshold=0.45)
    got2 = out2.to_lists()
    assert np.array_equal(ncand2.cpu().numpy(), ncand)
    for b in range(batch):
        for k in ('classes', 'scores', 'bboxes', 'anchor_index'):
            assert np.array_equal(got2[b][k], got[b][k]), (b, k)


@pytest.mark.parametrize('num_classes', CLASS_COUNTS)
def test_post_tfe_at_other_class_counts(dev, num_classes):
    from ron_tensorflow_amd import ops, tfe
    anchors = oanchors.anchors_all_layers()
    adev = ops.anchors_to_device(anchors, dev)
    batch, thr, nms, top_k, keep = 2, 0.01, 0.4, 200, 100
    cls, obj, loc = synth.head_tensors(400 + num_classes, batch=batch, num_classes=num_classes, bg=_bg(num_classes), ob=-2.0)
    loc = [l * np.float32(0.2) for l in loc]
    cls_d, obj_d, loc_d = _to_dev(cls, dev), _to_dev(obj, dev), _to_dev(loc, dev)
    pred = [ops.softmax_last(c) for c in cls_d]
    objp = [ops.softmax_last(o, pick=1) for o in obj_d]
    dec = [ops.bboxes_decode_layer(l, a) for l, a in zip(loc_d, adev)]
    gated = [(o > 0.03).to(torch.float32) * p for o, p in zip(objp, pred)]
    ds, db = tfe.detected_bboxes(gated, dec, num_classes=num_classes, select_threshold=thr, nms_threshold=nms,
                                 clipping_bbox=[0., 0., 1., 1.], top_k=top_k, keep_top_k=keep, nms_mode='min', min_size=0.03)
    rs, rb = tfe_post.detected_bboxes([g.cpu().numpy() for g in gated], [d.cpu().numpy() for d in dec], num_classes=num_classes,
                                      select_threshold=thr, nms_threshold=nms, clipping_bbox=[0., 0., 1., 1.],
                                      top_k=top_k, keep_top_k=keep, nms_mode='min', min_size=0.03)
    assert sorted(ds.keys()) == list(range(1, num_classes))
    n_kept = 0
    for c in range(1, num_classes):
        assert tuple(ds[c].shape) == (batch, keep)
        assert np.array_equal(ds[c].cpu().numpy(), rs[c]), c
        assert np.array_equal(db[c].cpu().numpy(), rb[c]), c
        n_kept += int((rs[c] > 0).sum())
    assert n_kept > 0


@pytest.mark.parametrize('num_classes', [2, 81])          # (21: tests/test_gpu_forward.py)
def test_forward_fp32_reducedfc_at_other_class_counts(dev, num_classes):
    import ron_tensorflow_amd.weights as W
    from ron_tensorflow_amd.nets import nets_factory
    weights = W.synthetic_weights('reducedfc', num_classes=num_classes, seed=11, bg=_bg(num_classes))
    images = W.synthetic_images(1, seed=4)
    pred, logits, objp, objl, loc, _ = orf.ron_forward(images, weights, 'reducedfc', num_classes=num_classes, backend='numpy')
    ron_class = nets_factory.get_network('ron_320_vgg')
    params = ron_class.default_params._replace(num_classes=num_classes)       # eval_ron_network.py:149
    net = ron_class(params, variant='reducedfc', dtype='fp32', max_batch=1).load_weights(weights)
    x = torch.from_numpy(images).to(dev)
    g_pred, g_logits, g_objp, g_objl, g_loc, _ = net.net(x, is_training=False)
    for i in range(4):
        assert tuple(g_logits[i].shape) == logits[i].shape and logits[i].shape[-1] == num_classes
        scale = np.abs(logits[i]).max()
        assert np.abs(g_logits[i].cpu().numpy() - logits[i]).max() < 1e-4 * scale, i
        assert np.abs(g_loc[i].cpu().numpy() - loc[i]).max() < 1e-4 * np.abs(loc[i]).max(), i
        np.testing.assert_allclose(g_pred[i].cpu().numpy(), pred[i], rtol=0, atol=1e-4)
    # the fused detect path == post-processing of the same context's heads
    anchors = oanchors.anchors_all_layers()
    det = net.detect(x).to_lists()
    ref = np_post.detect_from_predictions([p.cpu().numpy() for p in g_pred], [l.cpu().numpy() for l in g_loc], anchors,
                                          objness_pred=[o.cpu().numpy() for o in g_objp])
    assert np.array_equal(det[0]['classes'], ref[0]['classes']) and np.array_equal(det[0]['anchor_index'], ref[0]['anchor_index'])
    assert np.array_equal(det[0]['scores'], ref[0]['scores'])
    np.testing.assert_allclose(det[0]['bboxes'], ref[0]['bboxes'], rtol=0, atol=1e-5)
    net.close()


def test_class_count_limits(dev):
    from ron_tensorflow_amd import _lib
    from ron_tensorflow_amd.nets import nets_factory
    ron_class = nets_factory.get_network('ron_320_vgg')
    for bad in (1, 129):
        net = ron_class(ron_class.default_params._replace(num_classes=bad), variant='reducedfc', dtype='fp32', max_batch=1)
        with pytest.raises(_lib.RonError, match='num_classes'):
            net._context()                       # (the context is created on first use)
    net = ron_class(ron_class.default_params._replace(num_classes=128), variant='reducedfc', dtype='fp32', max_batch=1)
    assert net._context()
    net.close()
