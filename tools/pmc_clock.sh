#!/bin/bash
# Shader clock the conv kernels actually run at: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / 8 / dispatch duration.
# usage: tools/pmc_clock.sh <outdir> [layer,layer,...]
set -u
export TMPDIR=/tmp
# read by the HIP runtime when it initialises; under rocprofv3 that is before python starts (bench.py's setdefault comes too late there)
export GPU_MAX_HW_QUEUES=8
OUT=$1; L=${2:-conv4_2,b4_trio,fc6_full,conv3_2}
mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/clk -- python3 tools/sweep_conv.py --only $L --cfgs 0 --iters 6 > $OUT/clk.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(out + '/clk/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm_kernel' not in r['Kernel_Name']: continue
        d = rows[(f, r['Dispatch_Id'])]
        d[r['Counter_Name']] = float(r['Counter_Value'])
        d['grid'] = int(r['Grid_Size'])
        d['dur'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3 if 'End_Timestamp' in r else float('nan')
by_grid = collections.defaultdict(list)
for d in rows.values(): by_grid[d['grid']].append(d)
print('# grid(threads)  launches  avg_us  GRBM_GUI_ACTIVE/8  clock_GHz  MFMA_busy (of 1024 SIMD x cycles)')
for g in sorted(by_grid):
    v = by_grid[g]
    dur = sum(x['dur'] for x in v) / len(v)
    cyc = sum(x.get('GRBM_GUI_ACTIVE', 0) for x in v) / len(v) / 8
    mf = sum(x.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for x in v) / len(v)
    print('%9d %6d %9.1f %12.0f %8.3f %8.3f' % (g, len(v), dur, cyc, cyc / dur * 1e-3, mf / (1024 * cyc) if cyc else 0))
PY
