#!/bin/bash
# round 6, call: the tests that changed, the traffic of the default tile orders (panel model), post-processing regimes with clustered boxes
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_check3
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_gather_cabi.py tests/test_gpu_boundary.py tests/test_gpu_ron_eval.py tests/test_gpu_forward.py tests/test_gpu_benched_config.py -m gpu -q > $O/pytest_subset.txt 2>&1
tail -6 $O/pytest_subset.txt
bash tools/pmc_bench.sh $O/pmc_bf16 --in-flight 1 > $O/pmc.log 2>&1
cp $O/pmc_bf16/traffic_layers_full_bf16_bs32.txt $O/pmc_bf16/traffic_full_bf16_bs32.json $O/
rm -rf $O/pmc_bf16/pmc_fetch $O/pmc_bf16/pmc_write
cat $O/traffic_layers_full_bf16_bs32.txt
python3 tools/post_regimes.py > $O/post_regimes.txt 2>> $O/err.txt
cat $O/post_regimes.txt
for rep in 1 2; do
  RON_HIP_LIB=$PWD/tools/experiments/libron_hip_r05.so python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_prev_$rep.json 2>> $O/err.txt
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_this_$rep.json 2>> $O/err.txt
done
for f in bench_prev_1 bench_this_1 bench_prev_2 bench_this_2; do python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
    print(sys.argv[1], round(d['value'],1), 'ms', round(d['ms_per_step'],3))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 $O/err.txt
