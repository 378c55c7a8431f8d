#!/usr/bin/env python3
"""Reference point for the conv kernel's roofline: what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on this GPU for plain
bf16 GEMMs of the conv layers' implicit-GEMM shapes (no gather, no halo, no epilogue).  python tools/gemm_reference_point.py"""
import time

import torch

SHAPES = [('block4_trio3', 51200, 1536, 4608), ('conv4_2', 51200, 512, 4608), ('conv3_2', 204800, 256, 2304),
          ('fc6', 3200, 4096, 25088), ('block4_inc2_3x3', 51200, 1024, 9216), ('square 8192', 8192, 8192, 8192)]


def main():
    dev = torch.device('cuda:0')
    for name, m, n, k in SHAPES:
        a = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
        b = torch.randn((n, k), device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            c = a @ b.t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 10
        for _ in range(iters):
            c = a @ b.t()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print('%-18s M %6d N %5d K %6d  %8.1f us  %7.1f TFLOP/s' % (name, m, n, k, dt * 1e6, 2.0 * m * n * k / dt / 1e12), flush=True)
        del a, b, c


if __name__ == '__main__':
    main()
