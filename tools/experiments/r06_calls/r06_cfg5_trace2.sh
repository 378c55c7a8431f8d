#!/bin/bash
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_cfg5
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof5 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant ssd512 --batch 16 > $O/bench_cfg5_under_rocprof.json 2>> $O/err.txt
T=$(ls $O/prof5/*/*kernel_trace.csv | head -1)
gzip -c $T > $O/kernel_trace_cfg5.csv.gz
rm -rf $O/prof5
ls -la $O/kernel_trace_cfg5.csv.gz
