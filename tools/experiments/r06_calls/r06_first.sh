#!/bin/bash
# round 6, call 1: the default bench line with its sustained leg; the in-flight-2 kernel trace and its overlap account; the in-kernel
# K-loop clock inside the network (diagnostic library), after a burst and after seconds of load
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_first
mkdir -p $O
python3 bench.py --layers $O/layers_cfg2_inflight2.txt > $O/bench_cfg2_default.json 2> $O/bench_cfg2_default.err
tail -c 600 $O/bench_cfg2_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_under_rocprof_if2.json 2> $O/prof_if2.err
T=$(ls $O/prof_if2/*/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T --layers $O/layers_cfg2_inflight2.txt --skip-steps 4 --json $O/trace_overlap_if2.json > $O/trace_overlap_if2.txt 2>&1
cp $(ls $O/prof_if2/*/*kernel_stats.csv | head -1) $O/kernel_stats_if2.csv
gzip -c $T > $O/kernel_trace_if2.csv.gz
rm -rf $O/prof_if2
RON_HIP_LIB=$PWD/tools/experiments/libron_hip_stamps.so python3 tools/kloop_clock.py --json $O/kloop_clock_bf16_if2.json > $O/kloop_clock_bf16_if2.txt 2>&1
RON_HIP_LIB=$PWD/tools/experiments/libron_hip_stamps.so python3 tools/kloop_clock.py --in-flight 1 --json $O/kloop_clock_bf16_if1.json > $O/kloop_clock_bf16_if1.txt 2>&1
RON_HIP_LIB=$PWD/tools/experiments/libron_hip_stamps.so python3 tools/kloop_clock.py --dtype f16x3 --json $O/kloop_clock_f16x3_if2.json > $O/kloop_clock_f16x3_if2.txt 2>&1
cat $O/trace_overlap_if2.txt | head -40
cat $O/kloop_clock_bf16_if2.txt | head -40
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_first/bench_cfg2_default.json') if l.startswith('{')][-1])
print('value', d['value'], 'sustained', json.dumps(d.get('sustained'))[:900])
print('parity', json.dumps(d.get('parity_mode',{}).get('sustained'))[:700], d.get('parity_mode',{}).get('images_per_s'))
PY
