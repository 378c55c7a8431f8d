#!/bin/bash
# split-K finalize fused into the launch (last slice of a tile to arrive adds the slabs): parity, determinism, A/B against the separate pass
export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_forward.py tests/test_gpu_ssd.py tests/test_gpu_benched_config.py tests/test_gpu_num_classes.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python3 tools/soak_determinism.py 60 2>&1 | grep "runs differ"
for rep in 1 2; do for v in 1 0; do
  for b in 1 4; do
    RON_SPLITK_PASS=$v python3 bench.py --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('separate-pass $v ron batch $b', round(d['ms_per_step'],4))"
    RON_SPLITK_PASS=$v python3 bench.py --variant ssd512 --no-cpu-baseline --no-parity-mode --batch $b --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('separate-pass $v ssd batch $b', round(d['ms_per_step'],4))"
  done
  RON_SPLITK_PASS=$v python3 bench.py --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('separate-pass $v ron batch 32 default', round(d['value'],1), round(d['ms_per_step'],3))"
  RON_SPLITK_PASS=$v python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('separate-pass $v ron batch 32 in-flight 1', round(d['value'],1), round(d['ms_per_step'],3))"
  RON_SPLITK_PASS=$v python3 bench.py --variant ssd512 --batch 16 --no-cpu-baseline --no-parity-mode --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('separate-pass $v ssd batch 16 default', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
