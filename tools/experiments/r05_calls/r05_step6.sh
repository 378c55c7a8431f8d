#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
python3 tools/gemm_vs_vendor.py --rounds 3 > $O/gemm_vs_vendor_panel.txt 2> $O/step6.err
cat $O/gemm_vs_vendor_panel.txt
python3 tools/sweep_conv.py --cfgs=-1,0 --only fc6_full,fc7_full,b4_quad,b4_trio 2>> $O/step6.err
python3 tools/sweep_conv.py --batch 64 --cfgs=-1,0 --only fc6_full,fc7_full 2>> $O/step6.err
