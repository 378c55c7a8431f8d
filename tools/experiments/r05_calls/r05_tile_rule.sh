#!/bin/bash
# the 256 x 128 four-wave tile for launches of about one round of the chip (48 .. 320 tiles): parity, then A/B against the library before
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for lib in prevtile cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_$lib.so; fi
  for b in 1 2 4 8 16 32; do
    python3 bench.py --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib ron batch $b in-flight 1', round(d['ms_per_step'],4))"
  done
  python3 bench.py --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib ron batch 32 default', round(d['value'],1), round(d['ms_per_step'],3))"
  python3 bench.py --dtype f16x3 --no-cpu-baseline --no-parity-mode --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib ron f16x3 default', round(d['value'],1), round(d['ms_per_step'],3))"
  python3 bench.py --variant reducedfc --dtype fp16 --batch 64 --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg4', round(d['value'],1), round(d['ms_per_step'],3))"
  for b in 1 4 16; do
    python3 bench.py --variant ssd512 --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib ssd batch $b in-flight 1', round(d['ms_per_step'],4))"
  done
  python3 bench.py --variant ssd512 --batch 16 --no-cpu-baseline --no-parity-mode --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib ssd batch 16 default', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
