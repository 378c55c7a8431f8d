#!/bin/bash
# the picker's small-launch rule (128 x 64 for tiny launches of <= 256 columns): A/B at the small batches
export TMPDIR=/tmp
for rep in 1 2 3; do for lib in prevtile cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_$lib.so; fi
  for b in 1 2 4 8; do
    python3 bench.py --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib ron batch $b', round(d['ms_per_step'],4))"
  done
  for b in 1 4 16; do
    python3 bench.py --variant ssd512 --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib ssd batch $b', round(d['ms_per_step'],4))"
  done
done; done
