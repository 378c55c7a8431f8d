#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for lib in r04 cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_$lib.so; fi
  python3 bench.py --no-cpu-baseline --no-parity-mode --batch 4 --in-flight 1 --steps 200 --warmup 20 2>&1 | python3 -c "
import json,sys
L=[l for l in sys.stdin]
j=[l for l in L if l.startswith('{')]
print('batch 4 $lib', round(json.loads(j[-1])['ms_per_step'],4) if j else 'FAILED: '+''.join(L[-3:]))"
done; done
