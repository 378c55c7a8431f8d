#!/usr/bin/env python3
"""Experiment: the two execution slots of DetectPipeline on CU-masked streams (hipExtStreamCreateWithCUMask) - does giving each batch
its own part of the chip (own XCDs = own L2s) beat letting the two share all 256 CUs?   python tools/exp_cu_mask.py"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ron_tensorflow_amd import weights as W
from ron_tensorflow_amd.nets import nets_factory
from ron_tensorflow_amd.pipeline import DetectPipeline

hip = C.CDLL('libamdhip64.so')


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def run(pipe, x, steps=40, warm=8):
    pend = []
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        pend.append(pipe.submit(x))
        if len(pend) >= 2:
            pend.pop(0).wait()
    while pend:
        pend.pop(0).wait()
    torch.cuda.synchronize()
    return 32 * steps / (time.perf_counter() - t0)


def main():
    net = nets_factory.get_network('ron_320_vgg')(variant='full', dtype='bf16', max_batch=32, fuse_pools=True)
    net.load_weights(W.synthetic_weights('full', seed=1))
    x = torch.from_numpy(W.synthetic_images(32, seed=3)).cuda()
    pipe = DetectPipeline(net, slots=2)
    base = list(pipe.streams)
    masks = {
        'none': None,
        'halves contiguous (bits 0-127 / 128-255)': (range(0, 128), range(128, 256)),
        'halves by bit % 8 < 4 (if bits rotate over the XCDs: 4 XCDs each)': ([b for b in range(256) if b % 8 < 4], [b for b in range(256) if b % 8 >= 4]),
        'halves by bit % 2': ([b for b in range(256) if b % 2 == 0], [b for b in range(256) if b % 2 == 1]),
        'three quarters each, overlapping (0-191 / 64-255)': (range(0, 192), range(64, 256)),
    }
    for name, m in masks.items():
        pipe.streams = base if m is None else [masked_stream(m[0]), masked_stream(m[1])]
        for b, buf in enumerate(pipe.buffers):
            buf.record_stream(pipe.streams[b % 2])
        r = [run(pipe, x) for _ in range(3)]
        print('%-70s %s images/s' % (name, ' '.join('%.0f' % v for v in r)), flush=True)


if __name__ == '__main__':
    main()
