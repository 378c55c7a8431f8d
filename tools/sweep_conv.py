#!/usr/bin/env python3
"""Sweep the conv kernel's tile configurations over the RON-320 layer shapes (batch 32) on the GPU.

  python tools/sweep_conv.py [--dtype bf16] [--batch 32] [--cfgs 0,2,4] [--only name-substring]
Prints ms and TFLOP/s per (layer, cfg); used to choose conv_pick_cfg() (csrc/conv_mfma.hip)."""
import argparse
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ron_tensorflow_amd import _lib

# name, h, w, cin, cout, k, stride, rate, transpose
LAYERS = [
    ('conv1_2', 320, 320, 64, 64, 3, 1, 1, 0),
    ('conv2_1', 160, 160, 64, 128, 3, 1, 1, 0),
    ('conv2_2', 160, 160, 128, 128, 3, 1, 1, 0),
    ('conv3_1', 80, 80, 128, 256, 3, 1, 1, 0),
    ('conv3_2', 80, 80, 256, 256, 3, 1, 1, 0),
    ('conv4_1', 40, 40, 256, 512, 3, 1, 1, 0),
    ('conv4_2', 40, 40, 512, 512, 3, 1, 1, 0),
    ('conv5_1', 20, 20, 512, 512, 3, 1, 1, 0),
    ('fc6_full', 10, 10, 512, 4096, 7, 1, 1, 0),
    ('fc7_full', 10, 10, 4096, 4096, 1, 1, 1, 0),
    ('fc6_red', 10, 10, 512, 1024, 3, 1, 3, 0),
    ('b7_left_full', 10, 10, 4096, 512, 2, 2, 1, 0),
    ('b6_left_full', 10, 10, 4096, 512, 3, 1, 1, 0),
    ('b7_trio', 5, 5, 512, 2048, 3, 1, 1, 0),
    ('b7_inc2', 5, 5, 1024, 1024, 3, 1, 1, 0),
    ('b7_cls', 5, 5, 1024, 210, 3, 1, 1, 0),
    ('b7_loc', 5, 5, 512, 40, 3, 1, 1, 0),
    ('b6_trio', 10, 10, 512, 2048, 3, 1, 1, 0),
    ('b6_inc2', 10, 10, 1024, 1024, 3, 1, 1, 0),
    ('b6_cls', 10, 10, 1024, 210, 3, 1, 1, 0),
    ('b6_obj', 10, 10, 512, 20, 3, 1, 1, 0),
    ('b5_deconv', 10, 10, 512, 512, 2, 2, 1, 1),
    ('b5_trio', 20, 20, 512, 2048, 3, 1, 1, 0),
    ('b5_inc2', 20, 20, 1024, 1024, 3, 1, 1, 0),
    ('b5_cls', 20, 20, 1024, 210, 3, 1, 1, 0),
    ('b5_loc', 20, 20, 512, 40, 3, 1, 1, 0),
    ('b4_deconv', 20, 20, 512, 512, 2, 2, 1, 1),
    ('b4_trio', 40, 40, 512, 2048, 3, 1, 1, 0),
    ('b4_inc2', 40, 40, 1024, 1024, 3, 1, 1, 0),
    ('b4_cls', 40, 40, 1024, 210, 3, 1, 1, 0),
    ('b4_loc', 40, 40, 512, 40, 3, 1, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--cfgs', default='')
    ap.add_argument('--only', default='')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--cstride', type=int, default=0)
    ap.add_argument('--coff', type=int, default=0)
    ap.add_argument('--splitk', type=int, default=-1)
    a = ap.parse_args()
    lib = _lib.lib()
    ncfg = lib.ron_conv_num_tile_cfgs()
    cfgs = [int(c) for c in a.cfgs.split(',')] if a.cfgs else list(range(ncfg))     # 100 = halo-patch 3x3 kernel, -1 = auto
    print('%-14s %8s %9s | ' % ('layer', 'GFLOP', 'M') + ' '.join('cfg%-2d us/TF   ' % c for c in cfgs))
    for (name, h, w, cin, cout, k, stride, rate, tr) in LAYERS:
        if a.only and a.only not in name:
            continue
        ho, wo = (h, w) if (tr or stride == 1) else (h // stride, w // stride)
        flop = 2.0 * a.batch * (h * w if tr else ho * wo) * k * k * cin * cout
        cells = []
        for cfg in cfgs:
            d = _lib.ConvDesc(a.batch, h, w, cin, cout, k, k, stride, rate, 1, tr, _lib.DTYPES[a.dtype], cfg, a.cstride, a.coff, 0, a.splitk)
            ms = C.c_float()
            rc = lib.ron_conv2d_bench(C.byref(d), 3, a.iters, C.byref(ms))
            if rc != 0:
                cells.append('   n/a        ')
                continue
            cells.append('%7.1f/%-6.0f' % (ms.value * 1e3, flop / (ms.value * 1e-3) / 1e12))
        print('%-14s %8.1f %9d | ' % (name, flop / 1e9, a.batch * (h * w if tr else ho * wo)) + ' '.join(cells), flush=True)


if __name__ == '__main__':
    main()
