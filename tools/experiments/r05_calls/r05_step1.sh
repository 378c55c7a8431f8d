#!/bin/bash
# Round 5, step 1: conv_igemm beside the vendor GEMM on identical shapes + the vendor kernel names
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05
mkdir -p $O
python3 tools/gemm_vs_vendor.py > $O/gemm_vs_vendor.txt 2> $O/gemm_vs_vendor.err
cat $O/gemm_vs_vendor.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gemm -- python3 tools/gemm_vs_vendor.py --rounds 1 --iters 3 > $O/gemm_under_rocprof.txt 2>> $O/gemm_vs_vendor.err
f=$(ls $O/prof_gemm/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/gemm_kernel_stats.csv
t=$(ls $O/prof_gemm/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && python3 - "$t" > $O/gemm_kernel_trace_summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''), r.get('SGPR_Count', ''))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(k, []).append(d)
for k, v in agg.items():
    print('%9.1f us x %3d  grid %s wg %s lds %s vgpr %s agpr %s sgpr %s  %s' % (sorted(v)[len(v)//2], len(v), k[1], k[2], k[3], k[4], k[5], k[6], k[0][:200]))
PY
cat $O/gemm_kernel_trace_summary.txt
rm -rf $O/prof_gemm
timeout 900 python3 tools/gemm_vs_vendor.py --tunable --rounds 3 > $O/gemm_vs_vendor_tunable.txt 2> $O/gemm_vs_vendor_tunable.err
cat $O/gemm_vs_vendor_tunable.txt
cp gpurun_out/tunableop_results*.csv $O/ 2>/dev/null
python3 bench.py --no-cpu-baseline --no-parity-mode > $O/bench_head_default.json 2> $O/bench_head.err
tail -c 600 $O/bench_head_default.json
