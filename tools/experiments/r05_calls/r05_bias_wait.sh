#!/bin/bash
# the epilogue's per-row `s_waitcnt vmcnt(0)` (bias loads first used inside the row branch) removed: A/B against the library before
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -2
for lib in prevbias cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_$lib.so; fi
  echo "== $lib"; python3 tools/sweep_conv.py --cfgs=-1 --only conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,fc6_full,b4_cls,b4_quad,conv5_1 2>&1 | tail -10
done
for rep in 1 2 3; do for lib in prevbias cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_$lib.so; fi
  python3 bench.py --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib default', round(d['value'],1), round(d['ms_per_step'],3))"
  python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib in-flight 1', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
