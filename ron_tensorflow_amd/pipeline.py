"""Several batches in flight: execution slots over one set of weights, one stream each.

Every launch of the conv stack ends with a partial round of workgroups (batch 32: 400 tiles on 256 CUs = 1.56
rounds; 800 tiles = 3.125 ...) and with prologue / epilogue phases in which the matrix cores idle.  A serving loop
hides both by keeping a second batch in flight: while one slot's launch drains, the other slot's launch fills the
free CUs.  The slots are ``ron_clone`` contexts (shared packed weights, own activations / scratch / head buffers),
each fed on its own HIP stream; results are handed back through events, nothing blocks the host."""
import os
import warnings

import torch

from . import ops

from . import _HIP_UP_AT_IMPORT, _HW_QUEUES_PRESET      # state when the package was imported


def _check_hw_queues(streams_needed):
    """One hardware queue per stream: slots + a consumer + RCCL's stream + the default stream are more than the HIP runtime's
    default of 4, and streams that share a queue serialise behind each other (measured: -1.8 % images/s).  The runtime reads
    GPU_MAX_HW_QUEUES when it initialises: the package sets it to 8 at import (ron_tensorflow_amd/__init__.py) unless the
    application exported a value of its own.  That cannot help when HIP was already initialised by then (e.g. torch.cuda used before
    the import): the import time state is recorded below and that case is reported here.  A profiler's preloaded library (rocprofv3)
    initialises HIP before python starts and is NOT visible from here - the environment reads 8 by then although the runtime took its
    default: export GPU_MAX_HW_QUEUES=8 in the shell that starts the profiler (tools/profile_round.sh, tools/pmc_*.sh do)."""
    try:
        have = int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        have = 4
    if have < streams_needed:
        warnings.warn('DetectPipeline: %d streams in use but GPU_MAX_HW_QUEUES=%s: streams will share hardware queues and serialise; '
                      'export GPU_MAX_HW_QUEUES=8 before the first HIP call' % (streams_needed, os.environ.get('GPU_MAX_HW_QUEUES', 'unset (4)')),
                      RuntimeWarning, stacklevel=3)
    elif _HIP_UP_AT_IMPORT and not _HW_QUEUES_PRESET:
        warnings.warn('DetectPipeline: HIP was initialised before ron_tensorflow_amd was imported and GPU_MAX_HW_QUEUES was not exported by '
                      'then: the runtime keeps its default of 4 hardware queues for %d streams; export GPU_MAX_HW_QUEUES=8 before the '
                      'first HIP call' % streams_needed, RuntimeWarning, stacklevel=3)


class Ticket(object):
    """One submitted batch.  ``wait()`` makes the caller's current stream wait for it and returns the detections: reads
    of them belong on that stream.  ``release()`` (optional) marks, on the current stream, the point after which the
    detections are no longer read.  The output set comes round again ``slots * buffers_per_slot`` submissions later; the
    slot then waits for the release point or, without one, for everything the consumer's stream held at that moment.  A
    ticket whose output set has been handed to a later batch is expired: ``wait()`` on it raises instead of returning
    another batch's records."""

    def __init__(self, detections, done):
        self.detections, self._done = detections, done
        self._consumer = None            # stream wait() was called on
        self._release_event = None
        self._expired = False

    def wait(self):
        if self._expired:
            raise RuntimeError('DetectPipeline: this ticket was never waited for before its output set went to a later batch; '
                               'consume tickets within slots * buffers_per_slot submissions')
        cur = torch.cuda.current_stream()
        cur.wait_event(self._done)
        self.detections.record_stream(cur)
        self._consumer = cur
        return self.detections

    def release(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._release_event = ev


class DetectPipeline(object):
    """Round-robin submission of batches to ``slots`` execution slots of ``net``.

    ``submit(images)`` returns a Ticket; the detections it carries live in one of the slot's ``buffers_per_slot``
    output sets (allocated once for ``net.max_batch`` images; smaller batches get leading views) and stay valid for the
    next ``slots * buffers_per_slot - 1`` submissions, so a consumer (host copy, RCCL gather) can run on its own stream
    without holding the slot's next batch back.  Reuse is safe by construction, see Ticket.

    Needs one hardware queue per stream: GPU_MAX_HW_QUEUES >= slots + 3, read by the HIP runtime when it initialises (the package
    sets 8 at import; an application that touches the GPU before importing it exports the variable itself)."""

    def __init__(self, net, slots=2, top_k=400, buffers_per_slot=2, max_queued=8):
        assert slots >= 1 and buffers_per_slot >= 1
        # max_queued: submit() blocks the HOST while that many submitted batches have not finished on the GPU (0 = never).  Nothing
        # makes a host that enqueues faster than the GPU executes stop by itself: it piles up thousands of launches and events, and the
        # runtime then stalls it in bursts - round 6, batch 32, two slots: the driver's 20 timed steps 8 440 / 8 476 images/s unbounded,
        # 8 566 / 8 588 / 8 564 with 4 / 8 / 16 batches queued at most, 8 432 with 32; over 4 s the 100-step windows spread 7 632 ... 8 801
        # unbounded and 8 531 ... 8 737 with 8 (tools/experiments/r06_calls/r06_ahead.sh).  Eight batches are 30 ms of work: the GPU never
        # runs dry, the host is never more than that ahead.
        self.max_queued = int(max_queued)
        self._slack = 4 if self.max_queued >= 4 else 1
        self._queued = []
        if slots > 1:
            _check_hw_queues(slots + 3)               # + consumer + RCCL + the default stream
        self.net, self.top_k, self.buffers_per_slot = net, top_k, buffers_per_slot
        self.slots = [net] + [net.clone() for _ in range(slots - 1)]
        with torch.cuda.device(net.device):
            self.streams = [torch.cuda.Stream(device=net.device) for _ in self.slots]
        self.ready = [torch.cuda.Event() for _ in self.slots]
        n_sets = len(self.slots) * buffers_per_slot
        self.buffers = [ops.DetectionBuffers(net.max_batch, top_k, net.device) for _ in range(n_sets)]
        for b, buf in enumerate(self.buffers):
            buf.record_stream(self.streams[b % len(self.slots)])
        self._tickets = [None] * n_sets
        self._next = 0

    def submit(self, images, **detect_args):
        if self.max_queued > 0 and len(self._queued) >= self.max_queued + self._slack - 1:
            # host flow control: wait until fewer than max_queued batches are unfinished (one wait per `_slack` submissions, so that
            # small batches - 0.7 ms steps - do not pay a host wake-up per step; in between up to max_queued + _slack - 1 are queued)
            del self._queued[:self._slack - 1]
            self._queued.pop(0).synchronize()
        b = self._next                                # output set; slot = b % slots
        self._next = (b + 1) % len(self.buffers)
        i = b % len(self.slots)
        out = self.buffers[b].narrow(images.shape[0])
        cur = torch.cuda.current_stream()
        self.ready[i].record(cur)                     # `images` are settled on the caller's stream
        s = self.streams[i]
        s.wait_event(self.ready[i])
        prev = self._tickets[b]
        if prev is not None:                          # the batch that used this output set last
            prev._expired = True
            if prev._release_event is not None:       # its reader said when it was done
                s.wait_event(prev._release_event)
            elif prev._consumer is not None:          # it did not: wait for everything its stream holds right now
                ev = torch.cuda.Event()
                ev.record(prev._consumer)
                s.wait_event(ev)
        with torch.cuda.stream(s):
            self.slots[i].detect(images, top_k=self.top_k, out=out, **detect_args)
            done = torch.cuda.Event()
            done.record(s)
        t = Ticket(out, done)
        self._tickets[b] = t
        if self.max_queued > 0:
            self._queued.append(done)
        return t

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        self.synchronize()
        for slot in self.slots[1:]:
            slot.close()
        self.slots = self.slots[:1]
