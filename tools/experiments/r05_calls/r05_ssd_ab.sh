#!/bin/bash
# SSD-512: the block4 / block7 heads as two launches each (a library built with kSsdPairedHeads = 0) vs one two-output launch, same box
export TMPDIR=/tmp
for rep in 1 2; do
for b in 1 4 16; do
for lib in unpaired cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_$lib.so; fi
  fl=1; [ $b = 16 ] && fl=2
  python3 bench.py --variant ssd512 --no-cpu-baseline --no-parity-mode --batch $b --in-flight $fl --steps 100 --warmup 20 2>&1 | python3 -c "
import json,sys
L=[l for l in sys.stdin]
j=[l for l in L if l.startswith('{')]
print('ssd512 batch $b in-flight $fl $lib', (round(json.loads(j[-1])['ms_per_step'],4), round(json.loads(j[-1])['value'],1)) if j else 'FAILED: '+''.join(L[-3:]))"
done; done; done
