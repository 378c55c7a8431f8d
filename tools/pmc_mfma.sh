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
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 "$@" > $OUT/pmc_mfma.log 2>&1
python3 - "$OUT" "$@" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
args = sys.argv[2:]
def opt(name, default):
    return args[args.index(name) + 1] if name in args else default
key = '%s_%s_bs%s' % (opt('--variant', 'full'), opt('--dtype', 'bf16'), opt('--batch', '32'))
per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
dur = collections.defaultdict(float)      # ns of the dispatches, from the counter pass's own timestamps (one row per counter: count once)
have_ts = False
for f in glob.glob('%s/pmc_mfma/*/*counter_collection.csv' % out):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('<')[0].split('(')[0].replace('void ', '')
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            n[k] += 1
            if r.get('Start_Timestamp') and r.get('End_Timestamp'):
                have_ts = True
                dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
tot = collections.defaultdict(float)
for k in per:
    for c, v in per[k].items():
        tot[c] += v
SIMDS = 256 * 4
NOMINAL_GHZ = 2.4
def frac(d):
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is busy, summed over the SIMDs; GRBM_GUI_ACTIVE: busy cycles summed over
    # the 8 XCDs (guide: effective clock = GRBM_GUI_ACTIVE / 8 / wall time) -> elapsed cycles of the dispatches = GRBM_GUI_ACTIVE / 8
    el = d.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    return d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (SIMDS * el) if el > 0 else None
def clock(d, ns):
    # GHz the dispatches ran at in THIS (profiled, serialised) run: elapsed cycles / their duration
    return d.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / ns if ns > 0 else None
def entry(d, ns, cnt):
    e = {'dispatches': cnt, 'mfma_busy_fraction': frac(d), 'counters': dict(d)}
    if have_ts and ns > 0:
        e['duration_us_per_dispatch'] = ns / 1e3 / max(cnt, 1)
        e['clock_ghz'] = clock(d, ns)
        # busy fraction x delivered clock / nominal clock = the fraction of the NOMINAL (2.4 GHz, 2.5 PFLOP/s) matrix peak these
        # dispatches occupied: comparable with TFLOP/s-by-time / 2500 of the same dispatches when every MFMA is algorithmic work
        e['mfma_busy_x_clock_over_nominal'] = e['mfma_busy_fraction'] * e['clock_ghz'] / NOMINAL_GHZ
    return e
res = {'command': 'bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 ' + ' '.join(args), 'commit': os.environ.get('RON_COMMIT'),
       'formula': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), all dispatches of the run (7 steps) summed; '
                  'clock_ghz = GRBM_GUI_ACTIVE / 8 / dispatch duration (timestamps of the same counter pass)',
       'whole_run': entry(tot, sum(dur.values()), sum(n.values())),
       'per_kernel': {k: entry(per[k], dur[k], n[k]) for k in sorted(per)}}
json.dump(res, open('%s/mfma_busy_%s.json' % (out, key), 'w'), indent=1)
print(json.dumps({'whole_run': {k: v for k, v in res['whole_run'].items() if k != 'counters'},
                  'per_kernel': {k: {a: b for a, b in v.items() if a != 'counters'} for k, v in res['per_kernel'].items()}}, indent=1))
PY
