"""Several batches in flight: execution slots over one set of weights, one stream each.

Every launch of the conv stack ends with a partial round of workgroups (batch 32: 400 tiles on 256 CUs = 1.56
rounds; 800 tiles = 3.125 ...) and with prologue / epilogue phases in which the matrix cores idle.  A serving loop
hides both by keeping a second batch in flight: while one slot's launch drains, the other slot's launch fills the
free CUs.  The slots are ``ron_clone`` contexts (shared packed weights, own activations / scratch / head buffers),
each fed on its own HIP stream; results are handed back through events, nothing blocks the host."""
import torch

from . import ops


class Ticket(object):
    """One submitted batch: ``wait()`` makes the caller's current stream wait for it, then returns the detections;
    ``release()`` (optional) marks, on the current stream, the point after which the detections are no longer read -
    without it the buffers are simply assumed free by the time they come round again."""

    def __init__(self, detections, done, on_release):
        self.detections, self._done, self._on_release = detections, done, on_release

    def wait(self):
        torch.cuda.current_stream().wait_event(self._done)
        return self.detections

    def release(self):
        self._on_release()


class DetectPipeline(object):
    """Round-robin submission of batches to ``slots`` execution slots of ``net``.

    ``submit(images)`` returns a Ticket; the detections it carries live in one of the slot's ``buffers_per_slot``
    output sets and stay valid for the next ``slots * buffers_per_slot - 1`` submissions, so a consumer (host copy,
    RCCL gather) can run on its own stream without holding the slot's next batch back."""

    def __init__(self, net, slots=2, top_k=400, buffers_per_slot=2):
        assert slots >= 1 and buffers_per_slot >= 1
        self.net, self.top_k, self.buffers_per_slot = net, top_k, buffers_per_slot
        self.slots = [net] + [net.clone() for _ in range(slots - 1)]
        with torch.cuda.device(net.device):
            self.streams = [torch.cuda.Stream(device=net.device) for _ in self.slots]
        self.ready = [torch.cuda.Event() for _ in self.slots]
        n_sets = len(self.slots) * buffers_per_slot
        self.buffers = [None] * n_sets
        self.consumed = [torch.cuda.Event() for _ in range(n_sets)]
        self._released = [False] * n_sets
        self._next = 0

    def submit(self, images, **detect_args):
        b = self._next                                # output set; slot = b % slots
        self._next = (b + 1) % len(self.buffers)
        i = b % len(self.slots)
        n = images.shape[0]
        if self.buffers[b] is None or self.buffers[b].n != n:
            self.buffers[b] = ops.DetectionBuffers(n, self.top_k, self.net.device)
        cur = torch.cuda.current_stream()
        self.ready[i].record(cur)                     # `images` are settled on the caller's stream
        s = self.streams[i]
        s.wait_event(self.ready[i])
        if self._released[b]:                         # whoever read this output set last has said when it was done
            s.wait_event(self.consumed[b])
        with torch.cuda.stream(s):
            self.slots[i].detect(images, top_k=self.top_k, out=self.buffers[b], **detect_args)
            done = torch.cuda.Event()
            done.record(s)
        self._released[b] = False
        return Ticket(self.buffers[b], done, lambda: self._release(b))

    def _release(self, b):
        self.consumed[b].record(torch.cuda.current_stream())
        self._released[b] = True

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        self.synchronize()
        for slot in self.slots[1:]:
            slot.close()
        self.slots = self.slots[:1]
