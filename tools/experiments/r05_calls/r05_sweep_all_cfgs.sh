#!/bin/bash
# every row-gather tile configuration against the picker's choice, every layer shape, batches 4 .. 64: where does conv_pick_cfg leave time?
export TMPDIR=/tmp
for b in 4 8 16 32 64; do
  echo "== batch $b"
  python3 tools/sweep_conv.py --batch $b --cfgs=-1,0,7,10,2,9,3 2>&1 | grep -v "amdgpu.ids"
done
