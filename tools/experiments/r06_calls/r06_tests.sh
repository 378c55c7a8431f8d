#!/bin/bash
# round 6, call 2: the whole GPU suite on the new library; bench A/B against the round-5 library (split-K slab-size change); batch sweep
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_tests
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
for rep in 1 2; do
  RON_HIP_LIB=$PWD/tools/experiments/libron_hip_r05.so python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_prev_$rep.json 2>> $O/err.txt
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_this_$rep.json 2>> $O/err.txt
done
BATCHES="1 2 4 8 16 32" bash tools/batch_sweep.sh > $O/batch_sweep.txt 2>> $O/err.txt
echo "# round-5 library" >> $O/batch_sweep.txt
RON_HIP_LIB=$PWD/tools/experiments/libron_hip_r05.so BATCHES="1 2 4 8 16 32" bash tools/batch_sweep.sh >> $O/batch_sweep.txt 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant ssd512 --batch 16 > $O/bench_cfg5.json 2>> $O/err.txt
RON_HIP_LIB=$PWD/tools/experiments/libron_hip_r05.so python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant ssd512 --batch 16 > $O/bench_cfg5_prev.json 2>> $O/err.txt
cat $O/batch_sweep.txt
for f in bench_prev_1 bench_this_1 bench_prev_2 bench_this_2 bench_cfg5 bench_cfg5_prev; do python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
    print(sys.argv[1], round(d['value'],1), 'ms', round(d['ms_per_step'],3))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -5 $O/err.txt
