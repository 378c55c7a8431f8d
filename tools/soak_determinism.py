#!/usr/bin/env python3
"""Run-to-run determinism soak: the same batch through the same context many times, every head tensor compared bit for bit with
the first run -- for each dtype / variant / batch size, one batch in flight and two (the second slot runs other data meanwhile).
A counted-wait bug in a kernel shows here long before a parity tolerance notices it.   python tools/soak_determinism.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ron_tensorflow_amd import weights as W
from ron_tensorflow_amd.nets import nets_factory


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device('cuda:0')
    bad = 0
    for variant, dtype, batch in (('full', 'bf16', 32), ('full', 'f16x3', 32), ('full', 'fp32', 8), ('reducedfc', 'fp16', 64),
                                  ('full', 'bf16', 1), ('full', 'bf16', 4), ('full', 'fp32', 2), ('ssd512', 'bf16', 16),
                                  ('full', 'bf16', 16), ('full', 'f16x3', 12)):      # the mid (13..23) and level (<= 12) launch plans
        ssd = variant == 'ssd512'
        cls = nets_factory.get_network('ssd_512_vgg' if ssd else 'ron_320_vgg')
        if ssd:
            net = cls(dtype=dtype, max_batch=batch, device=dev, fuse_pools=True)
            net.load_weights(W.ssd_synthetic_weights(seed=5))
        else:
            net = cls(variant=variant, dtype=dtype, max_batch=batch, device=dev, fuse_pools=True)
            net.load_weights(W.synthetic_weights(variant, seed=1))
        shape = net.params.img_shape
        x = torch.from_numpy(W.synthetic_images(batch, seed=3, img_shape=shape)).to(dev)
        y = torch.flip(x, dims=[0]).contiguous()
        other = net.clone()
        side = torch.cuda.Stream(device=dev)
        ref = None
        diffs = 0
        for it in range(iters):
            if it % 2 == 1:                       # every other run shares the GPU with the second slot on another stream
                with torch.cuda.stream(side):
                    other.forward_heads(y)
            heads = net.forward_heads(x)
            cur = [t.clone() for grp in heads if grp is not None for t in grp]
            if ref is None:
                ref = cur
            elif any(not torch.equal(a, b) for a, b in zip(cur, ref)):
                diffs += 1
        torch.cuda.synchronize()
        print('%-10s %-6s batch %2d: %d of %d runs differ from the first' % (variant, dtype, batch, diffs, iters - 1), flush=True)
        bad += diffs
        other.close()
        net.close()
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
