#!/usr/bin/env python3
"""Per-tile overhead and per-K-step time of the 256 x 256 tile: 1x1 convolutions with M = 65536, N = 256 (exactly 256 tiles: one per
CU, one round) over a range of K.  Slope of time against K steps = the K step, intercept = set-up + prologue + epilogue of a tile.
  python3 tools/k_sweep.py [--dtype bf16] [--n 256]      (RON_IGEMM256_V1=1: the eight-wave loop)"""
import argparse
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ron_tensorflow_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--rows', type=int, default=65536)
    a = ap.parse_args()
    lib = _lib.lib()
    chunk = 64 if a.dtype in ('bf16', 'fp16') else 32
    pts = []
    for k in (256, 512, 1024, 2048, 4096, 8192):
        d = _lib.ConvDesc(a.rows // 1024, 32, 32, k, a.n, 1, 1, 1, 1, 0, 0, _lib.DTYPES[a.dtype], 0, 0, 0, 0, 1, 0)
        ms = C.c_float()
        best = 1e9
        for _ in range(3):
            rc = lib.ron_conv2d_bench(C.byref(d), 3, 20, C.byref(ms))
            if rc != 0:
                print('K %d: error %s' % (k, lib.ron_last_error()))
                break
            best = min(best, ms.value * 1e3)
        pts.append((k // chunk, best))
        print('K %5d  steps %4d  %8.1f us  %7.1f TFLOP/s' % (k, k // chunk, best, 2.0 * a.rows * a.n * k / best / 1e6), flush=True)
    (s0, t0), (s1, t1) = pts[2], pts[-1]
    slope = (t1 - t0) / (s1 - s0)
    print('K step %.3f us, per-tile overhead %.1f us (from the %d- and %d-step points)' % (slope, t0 - slope * s0, s0, s1))


if __name__ == '__main__':
    main()
