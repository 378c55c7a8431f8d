#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
echo "== conv5_1-shaped (20x20 512->512, 100 tiles of 256^2): cfg 0 (256^2 asm) / 2 / 9, split-K -1 (auto), 1, 2, 3"
for sk in -1 1 2 3; do echo "splitk $sk"; python3 tools/sweep_conv.py --cfgs=0,2,9 --only conv5_1 --splitk $sk 2>/dev/null | grep -v "^layer"; done
echo "== b5_cls b6_cls etc auto"
python3 tools/sweep_conv.py --cfgs=-1,0,2,9 --only b5_cls,b5_left,b4_left,b5_inc2,b5_trio,conv2_2,conv2_1 2>/dev/null
