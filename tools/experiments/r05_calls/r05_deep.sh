#!/bin/bash
# four LDS stages (three tiles in flight) for launches below one round of the chip: parity with the form forced on, then A/Bs
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
RON_DEEP_STAGES=1 timeout 1500 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_forward.py tests/test_gpu_ssd.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do for v in 0 1; do
  for b in 1 4; do
    RON_DEEP_STAGES=$v python3 bench.py --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deep $v ron batch $b', round(d['ms_per_step'],4))"
    RON_DEEP_STAGES=$v python3 bench.py --variant ssd512 --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deep $v ssd batch $b', round(d['ms_per_step'],4))"
  done
  RON_DEEP_STAGES=$v python3 bench.py --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deep $v ron batch 32 default', round(d['value'],1), round(d['ms_per_step'],3))"
  RON_DEEP_STAGES=$v python3 bench.py --variant ssd512 --batch 16 --no-cpu-baseline --no-parity-mode --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deep $v ssd batch 16 default', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
for v in 0 1; do
  RON_DEEP_STAGES=$v python3 bench.py --variant ssd512 --batch 16 --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 20 --warmup 5 --layers $O/deep${v}_ssd_layers.txt > /dev/null 2>&1
  RON_DEEP_STAGES=$v python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 20 --warmup 5 --layers $O/deep${v}_ron_layers.txt > /dev/null 2>&1
done
paste <(awk 'NR>2{print $1, $3}' $O/deep0_ssd_layers.txt) <(awk 'NR>2{print $3}' $O/deep1_ssd_layers.txt) | awk '{printf "%-40s off %7.1f on %7.1f\n", $1, $2, $3}' | tail -20
paste <(awk 'NR>2{print $1, $3}' $O/deep0_ron_layers.txt) <(awk 'NR>2{print $3}' $O/deep1_ron_layers.txt) | awk '{printf "%-40s off %7.1f on %7.1f\n", $1, $2, $3}' | tail -14
