#!/bin/bash
# Whole-step matrix-core utilisation from PMC counters ("MFMA utilisation (rocprof)" of the north-star): one --pmc pass (no
# tracing domains) over bench.py --in-flight 1, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE per dispatch.
# usage: tools/pmc_mfma.sh <outdir> [bench.py args...]      -> <outdir>/mfma_busy_<variant>_<dtype>_bs<batch>.json
set -u
export TMPDIR=/tmp
# read by the HIP runtime when it initialises; under rocprofv3 that is before python starts (bench.py's setdefault comes too late there)
export GPU_MAX_HW_QUEUES=8
OUT=$1; shift
mkdir -p $OUT
rm -rf $OUT/pmc_mfma          # one run per call: never add another call's dispatches (see pmc_bench.sh)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --in-flight 1 "$@" > $OUT/pmc_mfma.log 2>&1
python3 - "$OUT" "$@" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
args = sys.argv[2:]
def opt(name, default):
    return args[args.index(name) + 1] if name in args else default
key = '%s_%s_bs%s' % (opt('--variant', 'full'), opt('--dtype', 'bf16'), opt('--batch', '32'))
per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob('%s/pmc_mfma/*/*counter_collection.csv' % out):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('<')[0].split('(')[0].replace('void ', '')
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            n[k] += 1
tot = collections.defaultdict(float)
for k in per:
    for c, v in per[k].items():
        tot[c] += v
SIMDS = 256 * 4
def frac(d):
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is busy, summed over the SIMDs; GRBM_GUI_ACTIVE: busy cycles summed over
    # the 8 XCDs (guide: effective clock = GRBM_GUI_ACTIVE / 8 / wall time) -> elapsed cycles of the dispatches = GRBM_GUI_ACTIVE / 8
    el = d.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    return d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (SIMDS * el) if el > 0 else None
res = {'command': 'bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --in-flight 1 ' + ' '.join(args), 'commit': os.environ.get('RON_COMMIT'),
       'formula': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), all dispatches of the run (7 steps) summed',
       'whole_run': {'counters': dict(tot), 'mfma_busy_fraction': frac(tot)},
       'per_kernel': {k: {'dispatches': n[k], 'mfma_busy_fraction': frac(per[k]), 'counters': dict(per[k])} for k in sorted(per)}}
json.dump(res, open('%s/mfma_busy_%s.json' % (out, key), 'w'), indent=1)
print(json.dumps({'whole_run_mfma_busy_fraction': res['whole_run']['mfma_busy_fraction'],
                  'per_kernel': {k: v['mfma_busy_fraction'] for k, v in res['per_kernel'].items()}}, indent=1))
PY
