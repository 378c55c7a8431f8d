#!/bin/bash
# persistent four-wave tiles (conv_igemm_persist.h): parity of the conv tests with the form forced on, then per-layer and whole-step A/B
export TMPDIR=/tmp
RON_IGEMM_PERSIST=1 timeout 1500 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -4
for v in 0 1; do
  echo "== RON_IGEMM_PERSIST=$v"
  RON_IGEMM_PERSIST=$v python3 tools/sweep_conv.py --cfgs=-1 --only conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,fc6_full,b4_cls 2>&1 | tail -8
done
for rep in 1 2; do for v in 0 1; do
  RON_IGEMM_PERSIST=$v python3 bench.py --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persist $v default', round(d['value'],1), round(d['ms_per_step'],3))"
  RON_IGEMM_PERSIST=$v python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persist $v in-flight 1', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
