#!/bin/bash
# round 6: multiplier divisions in the tile set-up (FastDiv) against the library without them (libron_hip_r06b.so): conv + forward parity,
# batch sweep one in flight, the default line, the per-launch table
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=${O:-gpurun_out/r06_fastdiv}
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_forward.py tests/test_gpu_benched_config.py tests/test_gpu_ssd.py -m gpu -q -x > $O/pytest_subset.txt 2>&1
tail -4 $O/pytest_subset.txt
PREV=${PREV:-$PWD/tools/experiments/libron_hip_r06b.so}
for rep in 1 2; do
  RON_HIP_LIB=$PREV python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_prev_$rep.json 2>> $O/err.txt
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_this_$rep.json 2>> $O/err.txt
done
RON_HIP_LIB=$PREV python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 --layers $O/layers_prev.txt > $O/bench_if1_prev.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 --layers $O/layers_this.txt > $O/bench_if1_this.json 2>> $O/err.txt
echo "# this library" > $O/batch_sweep.txt
BATCHES="1 2 4 8 16 32" bash tools/batch_sweep.sh >> $O/batch_sweep.txt 2>> $O/err.txt
echo "# without FastDiv (libron_hip_r06b.so)" >> $O/batch_sweep.txt
RON_HIP_LIB=$PREV BATCHES="1 2 4 8 16 32" bash tools/batch_sweep.sh >> $O/batch_sweep.txt 2>> $O/err.txt
echo "# this library, again" >> $O/batch_sweep.txt
BATCHES="1 2 4 8" bash tools/batch_sweep.sh >> $O/batch_sweep.txt 2>> $O/err.txt
cat $O/batch_sweep.txt
paste $O/layers_prev.txt $O/layers_this.txt | awk '{printf "%-34s %9s %9s\n", $1, $3, $8}'
for f in bench_prev_1 bench_this_1 bench_prev_2 bench_this_2 bench_if1_prev bench_if1_this; do python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
    print(sys.argv[1], round(d['value'],1), 'ms', round(d['ms_per_step'],3))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
