#!/bin/bash
# Round-2 first GPU pass: the whole -m gpu suite, the default bench (new cpu_baseline / per-kernel / agreement fields), and the
# single-rank torchrun runs that look for the cost of the RCCL gather path (HW-queue count, consumer stream).
set -u
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
python3 bench.py --layers $O/layers_default.txt > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --no-cpu-baseline --in-flight 1 --layers $O/layers_if1.txt > $O/bench_if1.json 2> $O/bench_if1.err
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --no-cpu-baseline --check-gather"
$TR > $O/bench_torchrun.json 2> $O/bench_torchrun.err
GPU_MAX_HW_QUEUES=8 $TR > $O/bench_torchrun_q8.json 2> $O/bench_torchrun_q8.err
GPU_MAX_HW_QUEUES=8 python3 bench.py --no-cpu-baseline > $O/bench_default_q8.json 2> $O/bench_default_q8.err
tail -3 $O/pytest_gpu.txt
for f in bench_default bench_if1 bench_torchrun bench_torchrun_q8 bench_default_q8; do python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
    r=d['roofline']
    print(sys.argv[1], round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'per_kernel', round(r.get('per_kernel_frac',0),3), d.get('gather_check'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
