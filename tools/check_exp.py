#!/usr/bin/env python3
"""Correctness of experimental tile configurations (libron_hip_exp.so) against the oracle conv, before one is promoted.
  python tools/check_exp.py 14,16,17"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['RON_HIP_LIB'] = os.path.join(ROOT, 'tools', 'experiments', 'libron_hip_exp.so')
import numpy as np
import torch
from oracle import ron_forward as orf
from ron_tensorflow_amd import ops

cfgs = [int(c) for c in sys.argv[1].split(',')]
dev = torch.device('cuda:0')
bad = 0
for cfg in cfgs:
    for dtype, rnd, eps in (('bf16', orf.round_bf16, 2 ** -7 * 1.5), ('fp32', lambda a: a, 3e-5)):
        for (n, h, w, cin, cout, k, st, splitk) in ((3, 13, 11, 128, 192, 3, 1, -1), (2, 9, 9, 64, 256, 1, 1, -1), (2, 10, 10, 128, 256, 3, 1, 1),
                                                    (2, 20, 20, 192, 512, 3, 1, 1), (2, 5, 5, 256, 256, 3, 1, 3), (2, 10, 10, 128, 256, 2, 2, 1), (1, 10, 10, 64, 256, 7, 1, 2)):
            rs = np.random.RandomState(cfg + n + h)
            x = rs.randn(n, h, w, cin).astype(np.float32)
            wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
            b = (rs.randn(cout) * 0.1).astype(np.float32)
            ref = np.maximum(orf.conv2d_np(rnd(x), rnd(wt), st) + b, 0)
            got = ops.conv2d_nhwc(torch.from_numpy(x).to(dev), wt, b, stride=st, relu=True, dtype=dtype, tile_cfg=cfg, splitk=splitk).cpu().numpy()
            err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-6)
            ok = err <= eps
            bad += not ok
            print('cfg %d %s %s splitk %d: err %.3g %s' % (cfg, dtype, (n, h, w, cin, cout, k, st), splitk, err, 'ok' if ok else 'FAIL'))
print('FAILED %d' % bad if bad else 'all ok')
sys.exit(1 if bad else 0)
