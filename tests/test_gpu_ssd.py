"""GPU parity: SSD-512 (BASELINE config 5) -- conv stack vs oracle/ssd_forward.py, detections vs the np_methods oracle."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import np_post, tfe_post  # noqa: E402
from oracle import ron_forward as orf  # noqa: E402
from oracle import ssd_forward as osf  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def setup():
    import ron_tensorflow_amd.weights as W
    from ron_tensorflow_amd.nets import nets_factory
    weights = W.ssd_synthetic_weights(seed=5)
    images = W.synthetic_images(1, seed=4, img_shape=(512, 512))
    col = {}
    ref = osf.ssd_forward(images, weights, collect=col)
    return dict(W=W, factory=nets_factory, weights=weights, images=images, ref=ref, col=col)


def _rel(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


def test_ssd_fp32_forward_and_detect(setup, dev):
    cls = setup['factory'].get_network('ssd_512_vgg')
    net = cls(cls.default_params._replace(num_classes=21), dtype='fp32', max_batch=1).load_weights(setup['weights'])
    assert net.variables() == [(n, tuple(s)) for n, s in setup['W'].ssd_variable_shapes()]
    assert abs(net.flops_per_image() / 1e9 - 180.4) < 0.6
    x = torch.from_numpy(setup['images']).to(dev)
    net.params = net.params._replace(feat_shapes=[(1, 1)] * 7)          # stale shapes ...
    pred, loc, logits, eps = net.net(x, is_training=False, update_feat_shapes=False)
    assert net.params.feat_shapes == [(1, 1)] * 7                       # ... stay when the caller says so,
    pred, loc, logits, eps = net.net(x, is_training=False)              # and follow the predictions by default
    assert net.params.feat_shapes == [[64, 64, 4], [32, 32, 6], [16, 16, 6], [8, 8, 6], [4, 4, 6], [2, 2, 4], [1, 1, 4]]   # nets/ssd_vgg_300.py:297
    r_pred, r_loc, r_logits, r_eps = setup['ref']
    assert len(pred) == 7 and sorted(eps) == sorted(['block4', 'block7', 'block8', 'block9', 'block10', 'block11', 'block12'])
    for i in range(7):
        assert tuple(logits[i].shape) == r_logits[i].shape
        assert _rel(logits[i].cpu().numpy(), r_logits[i]) < 1e-4, i
        assert _rel(loc[i].cpu().numpy(), r_loc[i]) < 1e-4, i
        np.testing.assert_allclose(pred[i].cpu().numpy(), r_pred[i], rtol=0, atol=1e-4)
    for name in ('block4', 'block7', 'block8', 'block12'):
        assert _rel(eps[name].cpu().numpy(), r_eps[name]) < 1e-4, name
    assert _rel(net.end_point('block4_norm', 1).cpu().numpy(), setup['col']['block4_norm']) < 1e-4
    # anchors through the class (host C function) == oracle
    for a, b in zip(net.anchors((512, 512)), osf.anchors_all_layers()):
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    # fused detect (np_methods semantics, no objectness gate) on the heads this context produced
    anchors = osf.anchors_all_layers()
    det = net.detect(x).to_lists()[0]
    want = np_post.detect_from_predictions([p.cpu().numpy() for p in pred], [l.cpu().numpy() for l in loc], anchors,
                                           objness_pred=None, prior_scaling=net.params.prior_scaling)[0]
    assert want['n_candidates'] > 400
    assert np.array_equal(det['classes'], want['classes'])
    assert np.array_equal(det['anchor_index'], want['anchor_index'])
    assert np.array_equal(det['scores'], want['scores'])
    np.testing.assert_allclose(det['bboxes'], want['bboxes'], rtol=0, atol=1e-5)
    # reference call order of eval_ssd_network.py:184-202: decode, then detected_bboxes (no clip, no size filter)
    dec = net.bboxes_decode(loc, net.anchors((512, 512)))
    ds, db = net.detected_bboxes(pred, dec, select_threshold=0.01, nms_threshold=0.45, clipping_bbox=None, top_k=400, keep_top_k=200)
    rs, rb = tfe_post.detected_bboxes([p.cpu().numpy() for p in pred], [d.cpu().numpy() for d in dec], num_classes=21,
                                      select_threshold=0.01, nms_threshold=0.45, clipping_bbox=None, top_k=400, keep_top_k=200,
                                      nms_mode='min', min_size=None)
    for c in range(1, 21):
        assert np.array_equal(ds[c].cpu().numpy(), rs[c]), c
        assert np.array_equal(db[c].cpu().numpy(), rb[c]), c
    assert net.grouped_launches() == 5 and sum(n.startswith('group[') for n in net.launch_plan()) == 5
    net.close()


def test_ssd_bf16_forward(setup, dev):
    ref = osf.ssd_forward(setup['images'], setup['weights'], round_fn=orf.round_bf16)
    cls = setup['factory'].get_network('ssd_512_vgg')
    net = cls(dtype='bf16', max_batch=1, fuse_pools=True).load_weights(setup['weights'])
    logits, _, loc = net.forward_heads(torch.from_numpy(setup['images']).to(dev))
    for i in range(7):
        assert _rel(logits[i].cpu().numpy(), ref[2][i]) < 0.05, (i, _rel(logits[i].cpu().numpy(), ref[2][i]))
        assert _rel(loc[i].cpu().numpy(), ref[1][i]) < 0.08, i
    net.close()


def test_ssd_split_precision_forward_and_detect(setup, dev):
    """SSD-512 in dtype 'f16x3' (split precision: fp32-grade results on the f16 matrix cores) against the all-fp32 oracle: head
    tensors within 2e-5 of their scale (rate-6 conv6, the stride-2 / 4x4 VALID extra blocks and the L2 normalisation included),
    detections of the fused path == the oracle's np_methods pipeline on the oracle's own heads up to threshold ties."""
    from ron_tensorflow_amd.metrics import detection_agreement
    cls = setup['factory'].get_network('ssd_512_vgg')
    net = cls(cls.default_params._replace(num_classes=21), dtype='f16x3', max_batch=1, fuse_pools=True).load_weights(setup['weights'])
    x = torch.from_numpy(setup['images']).to(dev)
    logits, _, loc = net.forward_heads(x)
    r_pred, r_loc, r_logits, _ = setup['ref']
    for i in range(7):
        assert _rel(logits[i].cpu().numpy(), r_logits[i]) < 2e-5, i
        assert _rel(loc[i].cpu().numpy(), r_loc[i]) < 2e-5, i
    assert _rel(net.end_point('block4_norm', 1).cpu().numpy(), setup['col']['block4_norm']) < 2e-5
    det = net.detect(x).to_lists()[0]
    want = np_post.detect_from_predictions(r_pred, r_loc, osf.anchors_all_layers(), objness_pred=None,
                                           prior_scaling=net.params.prior_scaling)[0]
    a = detection_agreement(det, want, tol=1e-4)
    assert a['reproduced'] >= 0.98 and a['within_tol'] == 1.0, a
    net.close()


def test_network_fn_ssd(setup, dev):
    fn = setup['factory'].get_network_fn('ssd_512_vgg', 21, is_training=False, weights=setup['weights'], dtype='bf16', max_batch=1)
    assert fn.default_image_size == 512
    out = fn(torch.from_numpy(setup['images']).to(dev), end_points=())
    assert len(out) == 4 and tuple(out[0][0].shape) == (1, 64, 64, 4, 21) and tuple(out[1][6].shape) == (1, 1, 1, 4, 4)
