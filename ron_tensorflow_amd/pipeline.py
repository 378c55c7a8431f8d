"""Several batches in flight: execution slots over one set of weights, one stream each.

Every launch of the conv stack ends with a partial round of workgroups (batch 32: 400 tiles on 256 CUs = 1.56
rounds; 800 tiles = 3.125 ...) and with prologue / epilogue phases in which the matrix cores idle.  A serving loop
hides both by keeping a second batch in flight: while one slot's launch drains, the other slot's launch fills the
free CUs.  The slots are ``ron_clone`` contexts (shared packed weights, own activations / scratch / head buffers),
each fed on its own HIP stream; results are handed back through events, nothing blocks the host."""
import torch

from . import ops


class Ticket(object):
    """One submitted batch: ``wait()`` makes the caller's current stream wait for it, then returns the detections."""

    def __init__(self, detections, done):
        self.detections, self._done = detections, done

    def wait(self):
        torch.cuda.current_stream().wait_event(self._done)
        return self.detections


class DetectPipeline(object):
    """Round-robin submission of batches to ``slots`` execution slots of ``net``.

    ``submit(images)`` returns a Ticket; the detections it carries live in the slot's buffers and stay valid until
    that slot is used again, i.e. for the next ``slots - 1`` submissions."""

    def __init__(self, net, slots=2, top_k=400):
        assert slots >= 1
        self.net, self.top_k = net, top_k
        self.slots = [net] + [net.clone() for _ in range(slots - 1)]
        with torch.cuda.device(net.device):
            self.streams = [torch.cuda.Stream(device=net.device) for _ in self.slots]
        self.ready = [torch.cuda.Event() for _ in self.slots]
        self.done = [torch.cuda.Event() for _ in self.slots]
        self.consumed = [None for _ in self.slots]
        self.buffers = [None for _ in self.slots]
        self._next = 0

    def submit(self, images, **detect_args):
        i = self._next
        self._next = (i + 1) % len(self.slots)
        n = images.shape[0]
        if self.buffers[i] is None or self.buffers[i].n != n:
            self.buffers[i] = ops.DetectionBuffers(n, self.top_k, self.net.device)
        cur = torch.cuda.current_stream()
        self.ready[i].record(cur)                     # `images` (and the slot's previous results) are settled on the caller's stream
        s = self.streams[i]
        s.wait_event(self.ready[i])
        with torch.cuda.stream(s):
            self.slots[i].detect(images, top_k=self.top_k, out=self.buffers[i], **detect_args)
            self.done[i].record(s)
        return Ticket(self.buffers[i], self.done[i])

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        self.synchronize()
        for slot in self.slots[1:]:
            slot.close()
        self.slots = self.slots[:1]
