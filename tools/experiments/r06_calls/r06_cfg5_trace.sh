#!/bin/bash
# round 6: the overlap account (tools/trace_overlap.py) of config 5 (SSD-512, batch 16, two in flight) and config 4
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_cfg5
mkdir -p $O
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant ssd512 --batch 16 --in-flight 1 --layers $O/layers_cfg5_if1.txt > $O/bench_cfg5_if1.json 2>> $O/err.txt
rocprofv3 --kernel-trace --output-format csv -d $O/prof5 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant ssd512 --batch 16 > $O/bench_cfg5_under_rocprof.json 2>> $O/err.txt
T=$(ls $O/prof5/*/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T --layers $O/layers_cfg5_if1.txt --skip-steps 4 --json $O/trace_overlap_cfg5.json > $O/trace_overlap_cfg5.txt 2>&1
rm -rf $O/prof5
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant reducedfc --dtype fp16 --batch 64 --in-flight 1 --layers $O/layers_cfg4_if1.txt > $O/bench_cfg4_if1.json 2>> $O/err.txt
rocprofv3 --kernel-trace --output-format csv -d $O/prof4 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --variant reducedfc --dtype fp16 --batch 64 > $O/bench_cfg4_under_rocprof.json 2>> $O/err.txt
T=$(ls $O/prof4/*/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T --layers $O/layers_cfg4_if1.txt --skip-steps 4 --json $O/trace_overlap_cfg4.json > $O/trace_overlap_cfg4.txt 2>&1
rm -rf $O/prof4
cat $O/trace_overlap_cfg5.txt; head -6 $O/trace_overlap_cfg4.txt; tail -2 $O/trace_overlap_cfg4.txt
