#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for b in 1 4; do
for lib in r04 cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_r04.so; fi
  python3 bench.py --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch $b $lib', round(d['ms_per_step'],4))"
done; done; done
