#!/usr/bin/env python3
"""Experiment: the batch as P independent sub-batch pipelines on P streams (P ron_ctx of max_batch B/P) vs one
pipeline of B.  Hypothesis: the backbone launches yield 0.78-0.81 full rounds of workgroups per CU (tile quantisation);
with two pipelines in flight the tail of one launch is filled by the other pipeline's next launch.

  python tools/exp_two_pipes.py [--batch 32] [--pipes 1,2,4] [--steps 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ron_tensorflow_amd import weights as W
from ron_tensorflow_amd.nets import nets_factory


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--pipes', default='1,2,4')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--variant', default='full')
    ap.add_argument('--multi-stream', action='store_true')
    ap.add_argument('--sub', type=int, default=0, help='images per pipeline (default batch / pipes); > batch/pipes = whole batches in flight')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    w = W.synthetic_weights(a.variant, seed=1)
    x = torch.from_numpy(W.synthetic_images(max(a.batch, 4 * a.sub), seed=5)).to(dev)
    cls = nets_factory.get_network('ron_320_vgg')
    for p in [int(v) for v in a.pipes.split(',')]:
        sub = a.sub if a.sub else a.batch // p
        nets = [cls(variant=a.variant, dtype='bf16', max_batch=sub, fuse_pools=True, multi_stream=a.multi_stream) for _ in range(p)]
        for n in nets:
            n.load_weights(w)
        streams = [torch.cuda.Stream(device=dev) for _ in range(p)]
        xs = [x[i * sub:(i + 1) * sub].contiguous() for i in range(p)]

        def step():
            outs = []
            for n, s, xi in zip(nets, streams, xs):
                with torch.cuda.stream(s):
                    outs.append(n.detect(xi))
            return outs
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print('pipes %d x batch %d: %.3f ms/step  %.0f images/s' % (p, sub, dt * 1e3, p * sub / dt), flush=True)
        for n in nets:
            n.close()


if __name__ == '__main__':
    main()
