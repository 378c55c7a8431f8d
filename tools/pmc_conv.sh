#!/bin/bash
# PMC passes over one conv layer/config of tools/sweep_conv.py (each counter group its own run).
# usage: tools/pmc_conv.sh <layer-substring> <cfg> <outdir>
set -u
export TMPDIR=/tmp
# read by the HIP runtime when it initialises; under rocprofv3 that is before python starts (bench.py's setdefault comes too late there)
export GPU_MAX_HW_QUEUES=8
L=$1; CFG=$2; OUT=$3
mkdir -p $OUT
rm -rf $OUT/p[0-9]*          # one call = one set of runs (never add a previous call's dispatches)
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/sweep_conv.py --only $L --cfgs $CFG --iters 3 > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if not any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv3x3_patch", "conv3x3_c64")): continue
        agg[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
for k in sorted(agg):
    v = agg[k]['v']
    print('%-28s n=%3d mean=%.6g' % (k, len(v), sum(v) / len(v)))
PY
