#!/bin/bash
# round 6: the pipeline's host flow control (max_queued) - default line again, both legs, and the test
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_queued
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_benched_config.py tests/test_gpu_torchrun.py -m gpu -q > $O/pytest_subset.txt 2>&1
tail -4 $O/pytest_subset.txt
for Q in 8 0 6 8 0 12; do
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 3 --max-queued $Q > $O/bench_Q$Q.json 2>> $O/err.txt
  python3 - "$O/bench_Q$Q.json" $Q <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); s=d['sustained']; w=s['window_images_per_s']
print('max_queued', sys.argv[2], 'timed', round(d['value'],1), 'sustained', round(s['images_per_s'],1), 'windows min/med/max', round(w['min']), round(w['median']), round(w['max']))
PY
done
python3 bench.py --no-cpu-baseline --sustained-seconds 3 --dtype f16x3 --max-queued 8 > $O/bench_f16x3_Q8.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --sustained-seconds 3 --dtype f16x3 --max-queued 0 > $O/bench_f16x3_Q0.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 3 --variant ssd512 --batch 16 --max-queued 8 > $O/bench_cfg5_Q8.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 3 --variant ssd512 --batch 16 --max-queued 0 > $O/bench_cfg5_Q0.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 3 --batch 1 --in-flight 1 --steps 200 --warmup 20 --max-queued 8 > $O/bench_b1_Q8.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 3 --batch 1 --in-flight 1 --steps 200 --warmup 20 --max-queued 0 > $O/bench_b1_Q0.json 2>> $O/err.txt
for f in bench_f16x3_Q8 bench_f16x3_Q0 bench_cfg5_Q8 bench_cfg5_Q0 bench_b1_Q8 bench_b1_Q0; do python3 - "$O/$f.json" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); s=d['sustained']
print(sys.argv[1], 'timed', round(d['value'],1), 'sustained', round(s['images_per_s'],1))
PY
done
