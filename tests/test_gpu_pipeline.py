"""GPU: execution slots (ron_clone) and several batches in flight give the results of the plain path, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(max_batch=4):
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.nets import nets_factory
    net = nets_factory.get_network('ron_320_vgg')(variant='reducedfc', dtype='bf16', max_batch=max_batch, fuse_pools=True)
    net.load_weights(W.synthetic_weights('reducedfc', seed=1))
    return net


def _same(a, b):
    for k in ('count', 'classes', 'scores', 'bboxes', 'anchor_index'):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


def test_clone_matches_owner_and_lifetime():
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd._lib import RonError
    net = _net()
    x = torch.from_numpy(W.synthetic_images(4, seed=9)).cuda()
    ref = net.detect(x)
    slot = net.clone()
    got = slot.detect(x)
    torch.cuda.synchronize()
    _same(ref, got)
    assert int(ref.count.sum()) > 0
    # the owner cannot go while a slot borrows its weights (C ABI: RON_ERR_STATE)
    from ron_tensorflow_amd import _lib
    assert _lib.lib().ron_destroy(net._ctx) != 0
    slot.close()
    net._slots = []
    net.close()


def test_pipeline_three_slots_many_batches():
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.pipeline import DetectPipeline
    net = _net()
    batches = [torch.from_numpy(W.synthetic_images(4, seed=20 + i)).cuda() for i in range(7)]
    refs = []
    for b in batches:
        d = net.detect(b)
        refs.append({k: getattr(d, k).clone() for k in ('count', 'classes', 'scores', 'bboxes', 'anchor_index')})
    torch.cuda.synchronize()
    pipe = DetectPipeline(net, slots=3)
    tickets = []
    for i, b in enumerate(batches):
        tickets.append(pipe.submit(b))
        if len(tickets) == 3:                      # consume the oldest before its slot is reused
            j = i - 2
            d = tickets.pop(0).wait()
            for k, v in refs[j].items():
                assert torch.equal(getattr(d, k), v), (j, k)
    j = len(batches) - len(tickets)
    while tickets:
        d = tickets.pop(0).wait()
        for k, v in refs[j].items():
            assert torch.equal(getattr(d, k), v), (j, k)
        j += 1
    pipe.close()
    net.close()


def test_pack_detections_matches_torch_packing():
    from ron_tensorflow_amd import parallel
    from ron_tensorflow_amd import weights as W
    net = _net()
    det = net.detect(torch.from_numpy(W.synthetic_images(4, seed=31)).cuda())
    ref = parallel.pack_records(det.classes, det.scores, det.bboxes, det.anchor_index, det.count)
    got = parallel.pack_detections(det)
    assert torch.equal(ref, got)
    cl, sc, bb, ai, cnt = parallel.unpack_records(got)
    assert torch.equal(cnt, det.count) and torch.equal(cl, det.classes) and torch.equal(bb, det.bboxes)
    net.close()


def test_load_from_tf_checkpoint(tmp_path):
    """synthetic weights -> TensorFlow V2 checkpoint files -> RONNet.load_checkpoint: identical detections."""
    from ron_tensorflow_amd import checkpoint
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.nets import nets_factory
    w = W.synthetic_weights('reducedfc', seed=1)
    checkpoint.write_checkpoint(str(tmp_path / 'model.ckpt-7'), w)
    x = torch.from_numpy(W.synthetic_images(2, seed=4)).cuda()
    a = _net(2)
    ref = a.detect(x)
    b = nets_factory.get_network('ron_320_vgg')(variant='reducedfc', dtype='bf16', max_batch=2, fuse_pools=True)
    b.load_checkpoint(str(tmp_path))                       # directory: resolved through the `checkpoint` state file
    got = b.detect(x)
    torch.cuda.synchronize()
    _same(ref, got)
    a.close()
    b.close()
