"""GPU: the configuration bench.py measures, checked against the oracle's conv stack -- not only through properties.

BASELINE.json config 2 as benched: ``ron_net`` full VGG-16 (fc6 7x7x512 -> 4096, /root/reference/nets/ron_vgg_320.py:434-508,
:478), bf16, batch 32, pools fused into the conv epilogues, conv1_1 + conv1_2 + pool1 as the fused stem kernel, split-K heads,
two batches in flight through ``DetectPipeline(slots=2)`` exactly as bench.py submits them.  For images of that batch the oracle
runs the same network on the CPU (torch back-end, a second or two per image) with operands rounded to bf16 where the device
rounds them, and the head tensors and the end points pool1, block4, block6 (K = 25 088), block7 and block4_ref are compared.
Tolerances are those of test_gpu_forward.test_forward_reduced_precision (a few bf16 ulps of the tensor's scale: the device
folds BatchNorm into the weights before rounding them, the oracle rounds first).

The second test reports what SURVEY.md 8(d) asks for the precision actually benchmarked: how many detections of the fp32
oracle (conv stack + np_methods pipeline, no rounding anywhere) the bf16 path reproduces."""
import numpy as np
import pytest
import torch

from oracle import anchors as oanchors
from oracle import np_post
from oracle import ron_forward as orf
from parity_record import record

pytestmark = pytest.mark.gpu

BATCH = 32
# images of the batch the oracle recomputes: the first, one in the middle and the LAST one -- the ragged last tiles of every
# launch (block6 / block7: M = 3 200 = 12.5 tiles of 256; the 5 x 5 heads: M = 800) belong to the last image
CHECK = (0, 17, 31)

# Bounds = 2 x the error measured on MI355X (profiles/r03/parity_errors.json, same keys; fraction of the tensor's max
# magnitude).  A key without an entry falls back to the blanket tolerance of its dtype.
BLANKET = {'bf16': 0.04, 'fp16': 0.006, 'ssd_cls': 0.05, 'ssd_loc': 0.08}
BOUNDS = {
    'cfg2/block4': 0.012, 'cfg2/block4_ref': 0.016, 'cfg2/block6': 0.014, 'cfg2/block7': 0.016,
    'cfg2/block7_ref': 0.018, 'cfg2/cls0': 0.0083, 'cfg2/cls1': 0.0089, 'cfg2/cls2': 0.0077,
    'cfg2/cls3': 0.0092, 'cfg2/loc0': 0.019, 'cfg2/loc1': 0.017, 'cfg2/loc2': 0.015,
    'cfg2/loc3': 0.02, 'cfg2/obj0': 0.0086, 'cfg2/obj1': 0.01, 'cfg2/obj2': 0.0097,
    'cfg2/obj3': 0.011, 'cfg2/pool1': 0.0067, 'cfg4/block4': 0.0018, 'cfg4/block5_ref': 0.0023,
    'cfg4/block6': 0.002, 'cfg4/block7': 0.0018, 'cfg4/cls0': 0.00082, 'cfg4/cls1': 0.00097,
    'cfg4/cls2': 0.0012, 'cfg4/cls3': 0.0011, 'cfg4/loc0': 0.0022, 'cfg4/loc1': 0.0022,
    'cfg4/loc2': 0.0023, 'cfg4/loc3': 0.002, 'cfg4/obj0': 0.0011, 'cfg4/obj1': 0.0013,
    'cfg4/obj2': 0.0014, 'cfg4/obj3': 0.0012, 'cfg4/pool1': 0.00089, 'cfg5/cls0': 0.0017,
    'cfg5/cls1': 0.0081, 'cfg5/cls2': 0.009, 'cfg5/cls3': 0.012, 'cfg5/cls4': 0.011,
    'cfg5/cls5': 0.0061, 'cfg5/cls6': 0.0017, 'cfg5/loc0': 0.015, 'cfg5/loc1': 0.016,
    'cfg5/loc2': 0.027, 'cfg5/loc3': 0.014, 'cfg5/loc4': 0.014, 'cfg5/loc5': 0.019,
    'cfg5/loc6': 0.027,
}


def _rel(got, ref):
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


def _check(key, got, ref, blanket):
    """records the achieved error under `key` and asserts it against BOUNDS[key] (2 x the committed measurement)"""
    err = record(key, _rel(got, ref))
    assert err < BOUNDS.get(key, BLANKET[blanket]), (key, err, BOUNDS.get(key, BLANKET[blanket]))


@pytest.fixture(scope='module')
def run():
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.nets import nets_factory
    from ron_tensorflow_amd.pipeline import DetectPipeline
    weights = W.synthetic_weights('full', seed=1)                      # bench.py's weights and images
    images = W.synthetic_images(BATCH, seed=3)
    net = nets_factory.get_network('ron_320_vgg')(variant='full', dtype='bf16', max_batch=BATCH, fuse_pools=True)
    net.load_weights(weights)
    x = torch.from_numpy(images).cuda()
    pipe = DetectPipeline(net, slots=2, top_k=400)
    args = dict(objectness_thres=0.03, select_threshold=0.01, nms_threshold=0.45)
    tickets = [pipe.submit(x, **args) for _ in range(4)]               # both slots, both output sets of each
    dets = [t.wait().to_lists() for t in tickets]
    torch.cuda.synchronize()
    yield dict(net=net, pipe=pipe, x=x, images=images, weights=weights, dets=dets)
    pipe.close()
    net.close()


@pytest.fixture(scope='module')
def oracle_bf16(run):
    out = {}
    for i in CHECK:
        col = {}
        o = orf.ron_forward(run['images'][i:i + 1], run['weights'], 'full', backend='torch', round_fn=orf.round_bf16, collect=col)
        col.update(o[5])                                               # block1..block7 end points
        out[i] = (o, col)
    return out


def test_both_slots_agree(run):
    """The two execution slots run the same launches over the same weights: identical records."""
    for d in run['dets'][1:]:
        for a, b in zip(run['dets'][0], d):
            assert np.array_equal(a['classes'], b['classes']) and np.array_equal(a['anchor_index'], b['anchor_index'])
            assert np.array_equal(a['scores'], b['scores']) and np.array_equal(a['bboxes'], b['bboxes'])


@pytest.mark.parametrize('slot', [0, 1])
def test_end_points_vs_oracle(run, oracle_bf16, slot):
    """Activations the pipeline's slots hold after the batch vs the bf16-rounded oracle, per image."""
    s = run['pipe'].slots[slot]
    for name, key in (('pool1', 'pool1'), ('block4', 'block4'), ('block6', 'block6'), ('block7', 'block7'),
                      ('block4_ref', 'block4_ref'), ('block7_ref', 'block7_ref')):
        got = s.end_point(name, BATCH).cpu().numpy()
        for i in CHECK:
            _check('cfg2/%s' % name, got[i:i + 1], oracle_bf16[i][1][key], 'bf16')


def test_heads_vs_oracle(run, oracle_bf16):
    """Head tensors of the benched configuration (same context, same launches, full batch) vs the oracle."""
    cls, obj, loc = run['net'].forward_heads(run['x'])
    for i in CHECK:
        o = oracle_bf16[i][0]
        for l in range(4):
            _check('cfg2/cls%d' % l, cls[l][i:i + 1].cpu().numpy(), o[1][l], 'bf16')
            _check('cfg2/obj%d' % l, obj[l][i:i + 1].cpu().numpy(), o[3][l], 'bf16')
            _check('cfg2/loc%d' % l, loc[l][i:i + 1].cpu().numpy(), o[4][l], 'bf16')


def test_pipeline_detections_equal_oracle_post_on_own_heads(run):
    """ron_detect inside the pipeline == the oracle's np_methods pipeline on the head tensors of the same launches."""
    cls, obj, loc = run['net'].forward_heads(run['x'])
    anchors = oanchors.anchors_all_layers()
    for i in CHECK:
        ref = np_post.detect_from_logits([t[i:i + 1].cpu().numpy() for t in cls], [t[i:i + 1].cpu().numpy() for t in obj],
                                         [t[i:i + 1].cpu().numpy() for t in loc], anchors)[0]
        g = run['dets'][0][i]
        assert np.array_equal(g['classes'], ref['classes']) and np.array_equal(g['anchor_index'], ref['anchor_index'])
        assert np.abs(g['scores'] - ref['scores']).max() <= 1e-6
        assert np.abs(g['bboxes'] - ref['bboxes']).max() <= 1e-5


def test_bf16_vs_fp32_oracle_agreement(run):
    """Detections of the benched bf16 path vs the all-fp32 oracle (own conv stack + np_methods pipeline).  bf16 head
    tensors differ from fp32 ones by ~1 % of the logit scale, so scores move in the third decimal and candidates near
    a threshold or an IoU cut change sides: the rate is reported, with floors that catch a broken path."""
    from ron_tensorflow_amd.metrics import detection_agreement
    anchors = oanchors.anchors_all_layers()
    rates = []
    for i in CHECK:
        o = orf.ron_forward(run['images'][i:i + 1], run['weights'], 'full', backend='torch')
        ref = np_post.detect_from_predictions(o[0], o[4], anchors, objness_pred=o[2])[0]
        a = detection_agreement(run['dets'][0][i], ref, tol=1e-4)
        rates.append(a)
        record('cfg2/bf16_vs_fp32_oracle/not_reproduced', 1.0 - a['reproduced'])
        record('cfg2/bf16_vs_fp32_oracle/max_score_diff', a['max_score_diff'])
        record('cfg2/bf16_vs_fp32_oracle/max_box_diff', a['max_box_diff'])
        print('image %d: %s' % (i, a))
    # the bf16-rounded ORACLE reproduces 98 % of the fp32 oracle's detections on these images (max score diff 3e-3, box 9e-4)
    assert min(a['reproduced'] for a in rates) >= 0.90
    assert max(a['max_score_diff'] for a in rates) <= 0.02 and max(a['max_box_diff'] for a in rates) <= 0.01


# ---------------------------------------------------------------------------------------------------------------------
# north_star's tolerance clause at speed: the split-precision mode ('f16x3') at config 2's size
# ---------------------------------------------------------------------------------------------------------------------
def test_config2_f16x3_detections_within_1e4_of_fp32_oracle():
    """ron_net full, batch 32, dtype 'f16x3' (bench.py --dtype f16x3), two slots through DetectPipeline: for the first, a middle and
    the last image of the batch >= 98 % of the fp32 oracle's detections (own conv stack + np_methods, /root/reference/nets/
    ron_vgg_320.py:434-508 + nets/np_methods.py:23-242) are reproduced with the same class and anchor index, and ALL matched
    scores and boxes are within 1e-4; head tensors within 1.5e-5 of their scale."""
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.metrics import detection_agreement
    from ron_tensorflow_amd.nets import nets_factory
    from ron_tensorflow_amd.pipeline import DetectPipeline
    weights = W.synthetic_weights('full', seed=1)
    images = W.synthetic_images(BATCH, seed=3)
    net = nets_factory.get_network('ron_320_vgg')(variant='full', dtype='f16x3', max_batch=BATCH, fuse_pools=True)
    net.load_weights(weights)
    x = torch.from_numpy(images).cuda()
    pipe = DetectPipeline(net, slots=2, top_k=400)
    args = dict(objectness_thres=0.03, select_threshold=0.01, nms_threshold=0.45)
    dets = [t.wait().to_lists() for t in [pipe.submit(x, **args) for _ in range(2)]]
    torch.cuda.synchronize()
    for a, b in zip(dets[0], dets[1]):
        assert np.array_equal(a['anchor_index'], b['anchor_index']) and np.array_equal(a['scores'], b['scores'])
    cls, obj, loc = net.forward_heads(x)
    anchors = oanchors.anchors_all_layers()
    for i in CHECK:
        o = orf.ron_forward(images[i:i + 1], weights, 'full', backend='torch')
        for l in range(4):
            for nm, g, r in (('cls', cls[l], o[1][l]), ('obj', obj[l], o[3][l]), ('loc', loc[l], o[4][l])):
                assert record('cfg2_f16x3/%s%d' % (nm, l), _rel(g[i:i + 1].cpu().numpy(), r)) < 1.5e-5, (nm, l, i)      # measured <= 6.8e-6
        ref = np_post.detect_from_predictions(o[0], o[4], anchors, objness_pred=o[2])[0]
        a = detection_agreement(dets[0][i], ref, tol=1e-4)
        print('image %d: %s' % (i, a))
        record('cfg2_f16x3/not_reproduced', 1.0 - a['reproduced'])
        record('cfg2_f16x3/max_score_diff', a['max_score_diff'])
        record('cfg2_f16x3/max_box_diff', a['max_box_diff'])
        assert a['reproduced'] >= 0.98 and a['within_tol'] == 1.0, a
        assert a['max_score_diff'] <= 1e-4 and a['max_box_diff'] <= 1e-4, a
    pipe.close()
    net.close()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs 4 and 5 at the sizes bench.py runs them
# ---------------------------------------------------------------------------------------------------------------------
def test_config4_reducedfc_fp16_batch64_vs_oracle():
    """Config 4: ron_net_reducedfc (nets/ron_vgg_320.py:510-580), fp16, batch 64, fused pools / stem, grouped heads, two slots:
    head tensors and end points of two images vs the fp16-rounded oracle; records of both slots identical."""
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.nets import nets_factory
    from ron_tensorflow_amd.pipeline import DetectPipeline
    weights = W.synthetic_weights('reducedfc', seed=1)
    images = W.synthetic_images(64, seed=3)
    net = nets_factory.get_network('ron_320_vgg')(variant='reducedfc', dtype='fp16', max_batch=64, fuse_pools=True)
    net.load_weights(weights)
    x = torch.from_numpy(images).cuda()
    pipe = DetectPipeline(net, slots=2, top_k=400)
    dets = [t.wait().to_lists() for t in [pipe.submit(x) for _ in range(2)]]
    torch.cuda.synchronize()
    for a, b in zip(dets[0], dets[1]):
        assert np.array_equal(a['anchor_index'], b['anchor_index']) and np.array_equal(a['scores'], b['scores'])
    cls, obj, loc = net.forward_heads(x)
    for i in (5, 63):                       # 63 = the last image: the ragged last tiles
        col = {}
        o = orf.ron_forward(images[i:i + 1], weights, 'reducedfc', backend='torch', round_fn=orf.round_f16, collect=col)
        col.update(o[5])
        for l in range(4):
            _check('cfg4/cls%d' % l, cls[l][i:i + 1].cpu().numpy(), o[1][l], 'fp16')
            _check('cfg4/obj%d' % l, obj[l][i:i + 1].cpu().numpy(), o[3][l], 'fp16')
            _check('cfg4/loc%d' % l, loc[l][i:i + 1].cpu().numpy(), o[4][l], 'fp16')
        for name in ('pool1', 'block4', 'block6', 'block7', 'block5_ref'):
            _check('cfg4/%s' % name, net.end_point(name, 64).cpu().numpy()[i:i + 1], col[name], 'fp16')
    pipe.close()
    net.close()


def test_config5_ssd512_bf16_batch16_vs_oracle():
    """Config 5: SSD-512 (nets/ssd_vgg_512.py:364-460), bf16, batch 16, fused pools / stem, grouped box convolutions: head
    tensors of two images vs the bf16-rounded oracle, detections vs the oracle's np_methods pipeline on the same heads."""
    from oracle import ssd_forward as osf
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.nets import nets_factory
    weights = W.ssd_synthetic_weights(seed=5)
    images = W.synthetic_images(16, seed=3, img_shape=(512, 512))
    cls_ = nets_factory.get_network('ssd_512_vgg')
    net = cls_(cls_.default_params._replace(num_classes=21), dtype='bf16', max_batch=16, fuse_pools=True).load_weights(weights)
    x = torch.from_numpy(images).cuda()
    logits, _, loc = net.forward_heads(x)
    det = net.detect(x, select_threshold=0.01, nms_threshold=0.45).to_lists()
    anchors = osf.anchors_all_layers()
    for i in (0, 15):                       # 15 = the last image
        ref = osf.ssd_forward(images[i:i + 1], weights, round_fn=orf.round_bf16, backend='torch')
        for l in range(7):
            _check('cfg5/cls%d' % l, logits[l][i:i + 1].cpu().numpy(), ref[2][l], 'ssd_cls')
            _check('cfg5/loc%d' % l, loc[l][i:i + 1].cpu().numpy(), ref[1][l], 'ssd_loc')
        # probabilities from the device softmax kernel (bit-identical to the one fused into the select kernel: numpy's may
        # differ in the last bit, which reorders near-ties among thousands of candidates)
        from ron_tensorflow_amd import ops
        pred = [ops.softmax_last(t[i:i + 1]).cpu().numpy() for t in logits]
        want = np_post.detect_from_predictions(pred, [t[i:i + 1].cpu().numpy() for t in loc], anchors, objness_pred=None,
                                               prior_scaling=net.params.prior_scaling)[0]
        g = det[i]
        assert np.array_equal(g['classes'], want['classes']) and np.array_equal(g['anchor_index'], want['anchor_index'])
        assert np.array_equal(g['scores'], want['scores'])
        assert np.abs(g['bboxes'] - want['bboxes']).max() <= 1e-5
        # ... and end to end on the oracle's OWN (numpy) softmax of those logits: the same detections up to the near-ties a last-bit
        # difference of a probability reorders among thousands of candidates
        pred_np = [np_post.softmax_last(t[i:i + 1].cpu().numpy()) for t in logits]
        e2e = np_post.detect_from_predictions(pred_np, [t[i:i + 1].cpu().numpy() for t in loc], anchors, objness_pred=None,
                                              prior_scaling=net.params.prior_scaling)[0]
        ka = set(zip(g['classes'].tolist(), g['anchor_index'].tolist()))
        kb = set(zip(e2e['classes'].tolist(), e2e['anchor_index'].tolist()))
        assert len(ka & kb) >= 0.99 * max(len(ka), len(kb), 1), (len(ka), len(kb), len(ka & kb))
        sa = dict(zip(zip(g['classes'].tolist(), g['anchor_index'].tolist()), g['scores'].tolist()))
        sb = dict(zip(zip(e2e['classes'].tolist(), e2e['anchor_index'].tolist()), e2e['scores'].tolist()))
        assert max(abs(sa[k] - sb[k]) for k in ka & kb) <= 1e-6
    net.close()
