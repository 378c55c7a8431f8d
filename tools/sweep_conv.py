#!/usr/bin/env python3
"""Sweep the conv kernel's tile configurations over the RON-320 layer shapes (batch 32) on the GPU.

  python tools/sweep_conv.py [--dtype bf16] [--batch 32] [--cfgs 0,2,4] [--only name-substring]
Prints us and TFLOP/s per (layer, cfg); used to choose conv_pick_cfg() / conv_patch_pick() (csrc/conv_mfma.hip, conv_patch.hip).
Configurations: csrc/conv_mfma.h kCfg* (0-3, 7, 9 row-gather tiles, 4-6 halo-patch N tiles, 8 resident-weight 3x3 on 64 channels); n/a = does not
cover that layer.  RON_HIP_LIB=<path> points the package at another build of the same ABI (an experiment build, tools/experiments/README.md)."""
import argparse
import ctypes as C
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if '--zeros' in sys.argv:       # all-zero operands: the clock-limited share of a kernel's time (csrc/ops.cpp, ron_conv2d_bench)
    sys.argv.remove('--zeros')
    os.environ['RON_BENCH_ZERO'] = '1'
from ron_tensorflow_amd import _lib

# name, h, w, cin, cout, k, stride, rate, transpose
LAYERS = [
    ('conv1_2', 320, 320, 64, 64, 3, 1, 1, 0),
    ('conv2_1', 160, 160, 64, 128, 3, 1, 1, 0),
    ('conv2_2', 160, 160, 128, 128, 3, 1, 1, 0),
    ('conv3_1', 80, 80, 128, 256, 3, 1, 1, 0),
    ('conv3_2', 80, 80, 256, 256, 3, 1, 1, 0),
    ('conv4_3', 40, 40, 512, 512, 3, 1, 1, 0),
    ('conv4_1', 40, 40, 256, 512, 3, 1, 1, 0),
    ('conv4_2', 40, 40, 512, 512, 3, 1, 1, 0),
    ('conv5_1', 20, 20, 512, 512, 3, 1, 1, 0),
    ('fc6_full', 10, 10, 512, 4096, 7, 1, 1, 0),
    ('fc7_full', 10, 10, 4096, 4096, 1, 1, 1, 0),
    ('fc6_red', 10, 10, 512, 1024, 3, 1, 3, 0),
    ('b7_left_full', 10, 10, 4096, 512, 2, 2, 1, 0),
    ('b6_left_full', 10, 10, 4096, 512, 3, 1, 1, 0),
    ('b7_trio', 5, 5, 512, 1536, 3, 1, 1, 0),
    ('b7_inc2', 5, 5, 1024, 1024, 3, 1, 1, 0),
    ('b7_cls', 5, 5, 1024, 210, 3, 1, 1, 0),
    ('b7_loc', 5, 5, 512, 40, 3, 1, 1, 0),
    ('b6_trio', 10, 10, 512, 1536, 3, 1, 1, 0),
    ('b6_inc2', 10, 10, 1024, 1024, 3, 1, 1, 0),
    ('b6_cls', 10, 10, 1024, 210, 3, 1, 1, 0),
    ('b6_obj', 10, 10, 512, 20, 3, 1, 1, 0),
    ('b5_deconv', 10, 10, 512, 512, 2, 2, 1, 1),
    ('b5_trio', 20, 20, 512, 1536, 3, 1, 1, 0),
    ('b5_inc2', 20, 20, 1024, 1024, 3, 1, 1, 0),
    ('b5_cls', 20, 20, 1024, 210, 3, 1, 1, 0),
    ('b5_loc', 20, 20, 512, 40, 3, 1, 1, 0),
    ('b4_deconv', 20, 20, 512, 512, 2, 2, 1, 1),
    ('b4_trio', 40, 40, 512, 1536, 3, 1, 1, 0),
    ('b4_inc2', 40, 40, 1024, 1024, 3, 1, 1, 0),
    ('b4_cls', 40, 40, 1024, 210, 3, 1, 1, 0),
    ('b4_loc', 40, 40, 512, 40, 3, 1, 1, 0),
    ('b4_obj', 40, 40, 512, 20, 3, 1, 1, 0),
    ('b4_left', 40, 40, 512, 512, 3, 1, 1, 0),
    ('b4_quad', 40, 40, 512, 2048, 3, 1, 1, 0),      # trio3 + the 1x1 inception branch in the centre tap: --center-from 1536
    ('b4_inc2x', 40, 40, 1024, 1024, 3, 1, 1, 0),     # inception-2 3x3 + 1x1: --center-from 512
    ('b4_inc2_3x3', 40, 40, 1024, 512, 3, 1, 1, 0),
    ('b4_inc2_1x1', 40, 40, 1024, 512, 1, 1, 1, 0),
    ('b4_inc1_1x1', 40, 40, 512, 512, 1, 1, 1, 0),
    ('b5_left', 20, 20, 512, 512, 3, 1, 1, 0),
    ('b5_cls420', 20, 20, 1024, 420, 3, 1, 1, 0),     # cls_pred with the real 2 x 10 x 21 outputs
    ('b4_cls420', 40, 40, 1024, 420, 3, 1, 1, 0),
    # Winograd F(2x2, 3x3) upper bound: its 16 transform-domain GEMMs of a 40 x 40 layer at batch 32 are each
    # M = 32 * 20 * 20 tiles, K = Cin, N = Cout; as ONE launch they have the GEMM shape of this 1x1 conv at --batch 128
    # (M = 16 * 12 800 rows), transforms free.  `--batch 128 --only wino_`
    # SSD-512's extra blocks at batch 16 (`--batch 16`): latency chains, a few tiles each (stride-2 3x3 layers as stride-1 layers of the
    # same GEMM shape: M = 16 x 16 / 16 x 4 rows)
    ('ssd_b10_3x3', 4, 4, 128, 256, 3, 1, 1, 0),
    ('ssd_b11_3x3', 2, 2, 128, 256, 3, 1, 1, 0),
    ('ssd_b9_3x3', 8, 8, 128, 256, 3, 1, 1, 0),
    ('wino_conv4_2', 40, 40, 512, 512, 1, 1, 1, 0),
    ('wino_b4_trio', 40, 40, 512, 1536, 1, 1, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--cfgs', default='')
    ap.add_argument('--only', default='', help='comma-separated layer names (exact) or one substring')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--cstride', type=int, default=0)
    ap.add_argument('--coff', type=int, default=0)
    ap.add_argument('--splitk', type=int, default=-1)
    ap.add_argument('--center-from', type=int, default=0, help='output channels >= this run the centre tap only (timing: the weights are random everywhere)')
    a = ap.parse_args()
    lib = _lib.lib()
    ncfg = lib.ron_conv_num_tile_cfgs()
    cfgs = [int(c) for c in a.cfgs.split(',')] if a.cfgs else [-1] + list(range(ncfg))     # -1 = conv_pick_cfg's choice
    print('%-14s %8s %9s | ' % ('layer', 'GFLOP', 'M') + ' '.join('cfg%-2d us/TF   ' % c for c in cfgs))
    for (name, h, w, cin, cout, k, stride, rate, tr) in LAYERS:
        if a.only and not (name in a.only.split(',') or (',' not in a.only and a.only in name)):
            continue
        ho, wo = (h, w) if (tr or stride == 1) else (h // stride, w // stride)
        flop = 2.0 * a.batch * (h * w if tr else ho * wo) * k * k * cin * cout
        cells = []
        for cfg in cfgs:
            d = _lib.ConvDesc(a.batch, h, w, cin, cout, k, k, stride, rate, 1, tr, _lib.DTYPES[a.dtype], cfg, a.cstride, a.coff, 0, a.splitk, a.center_from)
            ms = C.c_float()
            rc = lib.ron_conv2d_bench(C.byref(d), 3, a.iters, C.byref(ms))
            if rc != 0:
                cells.append('   n/a        ')
                continue
            cells.append('%7.1f/%-6.0f' % (ms.value * 1e3, flop / (ms.value * 1e-3) / 1e12))
        print('%-14s %8.1f %9d | ' % (name, flop / 1e9, a.batch * (h * w if tr else ho * wo)) + ' '.join(cells), flush=True)


if __name__ == '__main__':
    main()
