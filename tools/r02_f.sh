#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_forward.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 600 python3 tools/check_exp.py 14,16,17 > $O/check_exp.txt 2>&1
tail -4 $O/check_exp.txt
timeout 1500 python3 tools/sweep_conv.py --exp --only conv2_1,conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_1,fc6_full,fc7_full,b6_left_full,b6_trio,b5_trio,b5_inc2,b5_cls,b4_trio,b4_inc2,b4_cls,b4_left,b4_deconv --cfgs=-1,0,14,15,17,1,16 > $O/sweep_exp3.txt 2>&1
cat $O/sweep_exp3.txt
