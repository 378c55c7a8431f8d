#!/bin/bash
# split-K only where a cost model (chain shortened vs second launch + slab traffic) says it wins: A/B against the rule without the veto
export TMPDIR=/tmp
for rep in 1 2; do for v in 1 0; do
  if [ $v = 1 ]; then export RON_SPLITK_NO_VETO=1; else unset RON_SPLITK_NO_VETO; fi
  for b in 1 2 4 8; do
    python3 bench.py --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-veto $v ron batch $b', round(d['ms_per_step'],4))"
  done
  for b in 1 4; do
    python3 bench.py --variant ssd512 --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-veto $v ssd batch $b', round(d['ms_per_step'],4))"
  done
  python3 bench.py --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-veto $v ron batch 32 default', round(d['value'],1), round(d['ms_per_step'],3))"
  python3 bench.py --variant reducedfc --dtype fp16 --batch 64 --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-veto $v cfg4', round(d['value'],1), round(d['ms_per_step'],3))"
  python3 bench.py --variant ssd512 --batch 16 --no-cpu-baseline --no-parity-mode --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-veto $v ssd batch 16 default', round(d['value'],1), round(d['ms_per_step'],3))"
  python3 bench.py --variant ssd512 --batch 16 --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-veto $v ssd batch 16 in-flight 1', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
