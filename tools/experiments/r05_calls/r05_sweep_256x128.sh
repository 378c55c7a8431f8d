#!/bin/bash
# where does the 256 x 128 four-wave tile (cfg 10) beat the picker's choice?  every layer shape, batches 4 .. 64
export TMPDIR=/tmp
for b in 4 8 16 32 64; do
  echo "== batch $b"
  python3 tools/sweep_conv.py --batch $b --cfgs=-1,10 2>&1 | grep -v "amdgpu.ids\|^layer" 
done
