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


def test_slow_consumer_never_reads_overwritten_records():
    """A consumer stream that lags (a long kernel sits in front of each of its reads) and never calls release(): the
    slot must wait for it before it overwrites an output set -- reuse is safe by construction, not by convention."""
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.pipeline import DetectPipeline
    net = _net()
    fields = ('count', 'classes', 'scores', 'bboxes', 'anchor_index')
    batches = [torch.from_numpy(W.synthetic_images(4, seed=40 + i)).cuda() for i in range(6)]
    refs = []
    for b in batches:
        d = net.detect(b)
        refs.append({k: getattr(d, k).clone() for k in fields})
    torch.cuda.synchronize()
    assert any(not torch.equal(refs[0]['scores'], r['scores']) for r in refs[1:])
    pipe = DetectPipeline(net, slots=2, buffers_per_slot=1)          # an output set comes round after two submissions
    consumer = torch.cuda.Stream()
    copies = []
    for b in batches:
        t = pipe.submit(b)
        with torch.cuda.stream(consumer):
            d = t.wait()
            torch.cuda._sleep(40 * 1000 * 1000)                      # tens of ms: several forward passes long
            copies.append({k: getattr(d, k).clone() for k in fields})
    torch.cuda.synchronize()
    for i, (c, r) in enumerate(zip(copies, refs)):
        for k in fields:
            assert torch.equal(c[k], r[k]), (i, k)
    pipe.close()
    net.close()


def test_ticket_of_a_reused_output_set_fails_loudly():
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.pipeline import DetectPipeline
    net = _net()
    x = torch.from_numpy(W.synthetic_images(2, seed=50)).cuda()       # a batch below max_batch: leading views of the set
    pipe = DetectPipeline(net, slots=1, buffers_per_slot=1)
    t0 = pipe.submit(x)
    t1 = pipe.submit(x)
    with pytest.raises(RuntimeError):
        t0.wait()
    d = t1.wait()
    assert d.n == 2 and tuple(d.scores.shape) == (2, 400)
    torch.cuda.synchronize()
    ref = net.detect(x)
    torch.cuda.synchronize()
    _same(ref, d)
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


@pytest.mark.parametrize('max_queued', [0, 1, 3])
def test_host_flow_control_bounds_the_queue_and_changes_nothing(max_queued):
    """DetectPipeline.max_queued (round 6): submit() makes the HOST wait while that many submitted batches have not finished.  Same
    detections as the unbounded pipeline; never more than max_queued completion events outstanding; with a bound of 1 every batch but
    the newest has finished on the GPU when submit() returns."""
    from ron_tensorflow_amd import weights as W
    from ron_tensorflow_amd.pipeline import DetectPipeline
    net = _net()
    batches = [torch.from_numpy(W.synthetic_images(4, seed=40 + i)).cuda() for i in range(6)]
    refs = []
    for b in batches:
        d = net.detect(b)
        refs.append({k: getattr(d, k).clone() for k in ('count', 'classes', 'scores', 'bboxes', 'anchor_index')})
    torch.cuda.synchronize()
    pipe = DetectPipeline(net, slots=2, max_queued=max_queued)
    pending, done_events = [], []
    for i, b in enumerate(batches):
        t = pipe.submit(b)
        done_events.append(t._done)
        assert max_queued == 0 or len(pipe._queued) <= max_queued + pipe._slack - 1
        if max_queued == 1 and i >= 1:
            assert all(e.query() for e in done_events[:-1])          # everything but the batch just submitted has finished
        pending.append((i, t))
        if len(pending) == 2:
            j, tj = pending.pop(0)
            d = tj.wait()
            for k, v in refs[j].items():
                assert torch.equal(getattr(d, k), v), (j, k)
    for j, tj in pending:
        d = tj.wait()
        for k, v in refs[j].items():
            assert torch.equal(getattr(d, k), v), (j, k)
    torch.cuda.synchronize()
    pipe.close()
    net.close()
