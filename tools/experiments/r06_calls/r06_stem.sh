#!/bin/bash
# round 6: the fused stem with two tile rows per wave in its conv1_2 phase, against the library before (libron_hip_r06c.so)
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_stem
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_g8.py tests/test_gpu_pipeline.py tests/test_gpu_benched_config.py tests/test_gpu_ssd.py tests/test_gpu_full_size.py -m gpu -q -x > $O/pytest_subset.txt 2>&1
tail -4 $O/pytest_subset.txt
PREV=$PWD/tools/experiments/libron_hip_r06c.so
for rep in 1 2; do
  RON_HIP_LIB=$PREV python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_prev_$rep.json 2>> $O/err.txt
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_this_$rep.json 2>> $O/err.txt
done
RON_HIP_LIB=$PREV python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 --layers $O/layers_prev.txt > $O/bench_if1_prev.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 --layers $O/layers_this.txt > $O/bench_if1_this.json 2>> $O/err.txt
RON_HIP_LIB=$PREV python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant ssd512 --batch 16 --layers $O/layers5_prev.txt > $O/bench_cfg5_prev.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant ssd512 --batch 16 --layers $O/layers5_this.txt > $O/bench_cfg5_this.json 2>> $O/err.txt
RON_HIP_LIB=$PREV python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant reducedfc --dtype fp16 --batch 64 > $O/bench_cfg4_prev.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant reducedfc --dtype fp16 --batch 64 > $O/bench_cfg4_this.json 2>> $O/err.txt
head -4 $O/layers_prev.txt; head -4 $O/layers_this.txt | tail -1; head -4 $O/layers5_prev.txt | tail -1; head -4 $O/layers5_this.txt | tail -1
for f in bench_prev_1 bench_this_1 bench_prev_2 bench_this_2 bench_if1_prev bench_if1_this bench_cfg5_prev bench_cfg5_this bench_cfg4_prev bench_cfg4_this; do python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
    print(sys.argv[1], round(d['value'],1), 'ms', round(d['ms_per_step'],3))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
