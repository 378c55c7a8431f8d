#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_conv.py -m gpu -q -x > $O/pytest_conv.txt 2>&1
tail -4 $O/pytest_conv.txt
timeout 1500 python3 tools/sweep_conv.py --exp --only conv2_2,conv3_2,conv4_2,conv5_1,b5_trio,b5_inc2,b5_cls,b4_trio,b4_inc2,b4_cls,b4_loc,b4_left --cfgs=-1,0,1,4,5,6,8,9,10,11,12,13,14,15 > $O/sweep_exp2.txt 2>&1
cat $O/sweep_exp2.txt
