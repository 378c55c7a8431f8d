#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
timeout 1500 python3 tools/sweep_conv.py --exp --cfgs=-1,0,1,2,3,4,5,6,7,8,9,10,11,12,13 > $O/sweep_exp.txt 2>&1
cat $O/sweep_exp.txt
