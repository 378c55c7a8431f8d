#!/usr/bin/env python3
"""conv_igemm as a 1x1 convolution beside the vendor GEMM (torch.matmul -> hipBLASLt) on the same plain-GEMM shapes, in ONE process
with interleaved rounds (cdna_hip_programming.md rule 24): what the K loop of conv_mfma.hip gives away to the vendor's inner loop.

  python3 tools/gemm_vs_vendor.py [--rounds 5] [--iters 10] [--cfgs -1,0] [--tunable]
Operands: activations U[-1,1), weights U[-1,1)/sqrt(K) on both sides (ron_conv2d_bench's distributions); bf16.  --tunable enables
torch's TunableOp (every hipBLASLt / rocBLAS solution is timed, the fastest is used) - run it as a separate invocation."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if '--tunable' in sys.argv:
    os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '1'
    os.environ['PYTORCH_TUNABLEOP_TUNING'] = '1'
    os.environ.setdefault('PYTORCH_TUNABLEOP_FILENAME', os.path.join(ROOT, 'gpurun_out', 'tunableop_results.csv'))
    os.environ.setdefault('PYTORCH_TUNABLEOP_VERBOSE', '1')
import torch  # noqa: E402

from ron_tensorflow_amd import _lib  # noqa: E402

# name, (n, h, w) of the 1x1 convolution's input (M = n*h*w), N = cout, K = cin
SHAPES = [('square 8192', (8, 32, 32), 8192, 8192),
          ('square 4096', (4, 32, 32), 4096, 4096),
          ('M 8000 4096^2', (5, 40, 40), 4096, 4096),
          ('block4_trio3 GEMM', (32, 40, 40), 1536, 4608),
          ('block4_inc2 GEMM', (32, 40, 40), 1024, 9216),
          ('conv4_2 GEMM', (32, 40, 40), 512, 4608)]


def vendor_us(a, b, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b.t()
    e1.record()
    e1.synchronize()
    del c
    return e0.elapsed_time(e1) * 1e3 / iters


def ours_us(lib, shape, n, k, cfg, iters):
    d = _lib.ConvDesc(shape[0], shape[1], shape[2], k, n, 1, 1, 1, 1, 0, 0, _lib.DTYPES['bf16'], cfg, 0, 0, 0, -1, 0)
    ms = C.c_float()
    rc = lib.ron_conv2d_bench(C.byref(d), 2, iters, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else float('nan')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--cfgs', default='-1,0')
    ap.add_argument('--only', default='')
    ap.add_argument('--tunable', action='store_true')
    a = ap.parse_args()
    cfgs = [int(c) for c in a.cfgs.split(',')]
    lib = _lib.lib()
    dev = torch.device('cuda:0')
    print('# %s  torch %s  rounds %d x iters %d, median (min) us and TFLOP/s at the median; tunable=%s' %
          (torch.cuda.get_device_name(0), torch.__version__, a.rounds, a.iters, a.tunable))
    for name, shape, n, k in SHAPES:
        if a.only and a.only not in name:
            continue
        m = shape[0] * shape[1] * shape[2]
        g = torch.Generator(device=dev).manual_seed(7)
        A = (torch.rand((m, k), device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        B = ((torch.rand((n, k), device=dev, generator=g) * 2 - 1) / k ** 0.5).to(torch.bfloat16)
        for _ in range(3):
            vendor_us(A, B, 2)
        tv, to = [], {c: [] for c in cfgs}
        for _ in range(a.rounds):
            tv.append(vendor_us(A, B, a.iters))
            for c in cfgs:
                to[c].append(ours_us(lib, shape, n, k, c, a.iters))
        flop = 2.0 * m * n * k
        med = lambda v: sorted(v)[len(v) // 2]
        line = '%-18s M %6d N %5d K %5d | vendor %8.1f (%8.1f) us %7.1f TF' % (name, m, n, k, med(tv), min(tv), flop / med(tv) / 1e6)
        for c in cfgs:
            line += ' | conv_igemm cfg %2d %8.1f (%8.1f) us %7.1f TF' % (c, med(to[c]), min(to[c]), flop / med(to[c]) / 1e6)
        print(line, flush=True)
        del A, B


if __name__ == '__main__':
    main()
