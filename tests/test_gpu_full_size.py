"""GPU, BASELINE.json config 2 at its full size (ron_net full VGG-16, bf16, batch 32): the oracle's conv stack does not finish
such a batch in seconds, so parity is checked through size-independent properties of the path plus the oracle's
post-processing on the network's own head tensors for a few images."""
import numpy as np
import pytest
import torch

from oracle import anchors as oanchors
from oracle import np_post

pytestmark = pytest.mark.gpu

BATCH = 32


@pytest.fixture(scope='module')
def run():
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.nets import nets_factory
    net = nets_factory.get_network('ron_320_vgg')(variant='full', dtype='bf16', max_batch=BATCH, fuse_pools=True)
    net.load_weights(W.synthetic_weights('full', seed=1))
    x = torch.from_numpy(W.synthetic_images(BATCH, seed=3)).cuda()
    logits, objl, loc = net.forward_heads(x)
    det = net.detect(x)
    torch.cuda.synchronize()
    yield net, x, (logits, objl, loc), det
    net.close()


def test_record_invariants(run):
    _, _, _, det = run
    cnt = det.count.cpu().numpy()
    sc, cl, bb = det.scores.cpu().numpy(), det.classes.cpu().numpy(), det.bboxes.cpu().numpy()
    assert cnt.shape == (BATCH,) and (cnt > 0).all() and (cnt <= 400).all()
    for i in range(BATCH):
        k = cnt[i]
        assert (np.diff(sc[i, :k]) <= 0).all(), 'scores not descending'
        assert (sc[i, :k] > 0.01).all() and (cl[i, :k] >= 1).all() and (cl[i, :k] <= 20).all()
        assert not sc[i, k:].any() and not cl[i, k:].any() and not bb[i, k:].any(), 'rows past count must be zero'
        assert (bb[i, :k] >= 0).all() and (bb[i, :k] <= 1).all()              # clipped to bbox_img, resize is the identity


def test_nms_is_idempotent_and_leaves_no_overlaps(run):
    """sort + NMS applied to its own output changes nothing; no kept same-class pair reaches the IoU threshold."""
    from ron_tensorflow_amd import ops
    _, _, _, det = run
    again, _ = ops.np_sort_nms(det.classes, det.scores, det.bboxes, top_k=400, nms_threshold=0.45, n_valid=det.count)
    assert torch.equal(again.count, det.count)
    assert torch.equal(again.classes, det.classes) and torch.equal(again.scores, det.scores)
    assert torch.equal(again.bboxes, det.bboxes)
    lists = det.to_lists()
    for d in lists[:4]:
        for a in range(len(d['classes'])):
            same = np.nonzero(d['classes'][a + 1:] == d['classes'][a])[0] + a + 1
            if same.size:
                iou = np_post.bboxes_jaccard(d['bboxes'][a], d['bboxes'][same])
                assert (iou < 0.45).all()


def test_fused_detect_equals_oracle_on_own_heads(run):
    """ron_detect of the full batch == the oracle's np_methods pipeline on the head tensors ron_forward produced,
    image by image (ids and scores bit-exact, boxes 1e-5)."""
    _, _, (logits, objl, loc), det = run
    anchors = oanchors.anchors_all_layers()
    got = det.to_lists()
    for i in (0, 13, 31):
        ref = np_post.detect_from_logits([t[i:i + 1].cpu().numpy() for t in logits], [t[i:i + 1].cpu().numpy() for t in objl],
                                         [t[i:i + 1].cpu().numpy() for t in loc], anchors)[0]
        g = got[i]
        assert np.array_equal(g['classes'], ref['classes']) and np.array_equal(g['anchor_index'], ref['anchor_index'])
        assert np.abs(g['scores'] - ref['scores']).max() <= 1e-6
        assert np.abs(g['bboxes'] - ref['bboxes']).max() <= 1e-5


def test_batch_position_does_not_matter(run):
    """Image i of the batch of 32 == the same image fed as a batch of one: same tile configuration per launch only up
    to split-K, so head tensors agree to bf16 accumulation-order noise and the detections to the candidates that sit
    within that noise of a threshold."""
    net, x, (logits, _, _), det = run
    lg1, _, _ = net.forward_heads(x[7:8])
    for a, b in zip(lg1, logits):
        scale = float(b[7:8].abs().max())
        assert float((a - b[7:8]).abs().max()) <= 2e-2 * scale
    one = net.detect(x[7:8]).to_lists()[0]
    full = det.to_lists()[7]
    ka = set(zip(one['classes'].tolist(), one['anchor_index'].tolist()))
    kb = set(zip(full['classes'].tolist(), full['anchor_index'].tolist()))
    assert len(ka & kb) >= 0.95 * max(len(ka), len(kb))


def test_determinism_same_batch(run):
    net, x, _, det = run
    det2 = net.detect(x)
    for k in ('count', 'classes', 'scores', 'bboxes', 'anchor_index'):
        assert torch.equal(getattr(det, k), getattr(det2, k))
