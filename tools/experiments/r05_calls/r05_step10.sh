#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -12
echo "== bf16"; python3 tools/sweep_conv.py --cfgs=-1,1,9,10 --only conv2_2,conv2_1 2>/dev/null
echo "== f16x3"; python3 tools/sweep_conv.py --dtype f16x3 --cfgs=-1,1,9,10 --only conv2_2,conv2_1 2>/dev/null
echo "== ssd-size bf16 batch 16 (256x256 maps)"; python3 tools/sweep_conv.py --batch 64 --cfgs=-1,9,10 --only conv2_2 2>/dev/null
