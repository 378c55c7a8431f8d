#!/bin/bash
# kernel trace of config 5 (SSD-512, batch 16, one in flight): what the tail launches cost as kernels, without the events around them
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05; mkdir -p $O; rm -rf $O/prof_ssd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ssd -- python3 bench.py --variant ssd512 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --in-flight 1 > $O/ssd_under_rocprof.json 2> $O/ssd_trace_err.txt
f=$(ls $O/prof_ssd/*/*kernel_stats.csv | head -1); cp $f $O/ssd_kernel_stats_if1.csv
t=$(ls $O/prof_ssd/*/*kernel_trace.csv | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step: find the last stem kernel, print every kernel after it with duration and the gap to the previous one
idx = [i for i, r in enumerate(rows) if 'stem2' in r['Kernel_Name']]
s = idx[-2]; e = idx[-1]
prev_end = None
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('ron::detail::', '').replace('ron::(anonymous namespace)::', '')[:70]
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print('%-72s %8.1f us  gap %6.1f  grid %s' % (name, (en - st) / 1e3, gap, r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
    prev_end = en
print('step wall %.1f us' % ((int(rows[e]['Start_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1e3))
PY
rm -rf $O/prof_ssd
