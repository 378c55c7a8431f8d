#!/bin/bash
# Round 5, step 2: the four-wave assembly K loop: parity, then timing beside the eight-wave loop and the vendor GEMM
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -15 > $O/step2_pytest_conv.txt
cat $O/step2_pytest_conv.txt
python3 tools/gemm_vs_vendor.py --rounds 3 > $O/gemm_vs_vendor_4w.txt 2> $O/step2.err
cat $O/gemm_vs_vendor_4w.txt
L="conv3_2,conv4_2,b4_trio,b4_inc2,b4_quad,fc6_full,fc7_full,b4_cls,conv4_1,conv4_3,conv3_1"
python3 tools/sweep_conv.py --cfgs=-1,0,7 --only $L > $O/sweep_4w.txt 2>> $O/step2.err
RON_IGEMM256_V1=1 python3 tools/sweep_conv.py --cfgs=-1,0,7 --only $L > $O/sweep_8w.txt 2>> $O/step2.err
echo "== 4 waves (asm)"; cat $O/sweep_4w.txt; echo "== 8 waves"; cat $O/sweep_8w.txt
tail -5 $O/step2.err
