#!/bin/bash
# HBM traffic of the conv kernel from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (no tracing domains), FETCH_SIZE doubled on gfx950
# (it tallies 128-B requests at 64 B for wide coalesced streams), both counters are in KiB.
# usage: tools/pmc_bench.sh <outdir> [bench.py args...]      -> <outdir>/traffic_<variant>_<dtype>_bs<batch>.json
#                                                             + <outdir>/traffic_layers_<variant>_<dtype>_bs<batch>.txt (per launch)
# The counter directories are removed first: a second call with other bench.py args into the same <outdir> must not add its
# dispatches to the first call's (round 3's f16x3 files summed the bf16 and the f16x3 passes that way).
set -u
export TMPDIR=/tmp
# read by the HIP runtime when it initialises; under rocprofv3 that is before python starts (bench.py's setdefault comes too late there)
export GPU_MAX_HW_QUEUES=8
OUT=$1; shift
mkdir -p $OUT
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_layers.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --layers $OUT/pmc_layers.txt "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 "$@" > $OUT/pmc_write.log 2>&1
python3 - "$OUT" "$@" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
args = sys.argv[2:]
def opt(name, default):
    return args[args.index(name) + 1] if name in args else default
key = '%s_%s_bs%s' % (opt('--variant', 'full'), opt('--dtype', 'bf16'), opt('--batch', '32'))
CONV = ('conv_igemm', 'conv3x3_patch', 'conv3x3_c64')             # every conv launch, grouped ones included (the headline set)
STEM = ('stem2_kernel', 'stem_conv_kernel', 'stem_conv_split_kernel')                        # the fused stem: a row of the per-launch table, not of the headline
tot, seq = {}, {}
for which in ('fetch', 'write'):
    rows = []
    for f in glob.glob('%s/pmc_%s/*/*counter_collection.csv' % (out, which)):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    rows.sort()
    vals = [v for _, k, v in rows if any(c in k for c in CONV)]
    tot[which] = (sum(vals) / max(len(vals), 1), len(vals))
    seq[which] = [(k, v) for _, k, v in rows if any(c in k for c in CONV + STEM)]
fetch_kib, n1 = tot['fetch']
write_kib, n2 = tot['write']
res = {'command': 'bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 ' + ' '.join(args), 'commit': os.environ.get('RON_COMMIT'),
       'conv_launches_sampled': [n1, n2],
       'FETCH_SIZE_KiB_per_launch_raw': fetch_kib, 'WRITE_SIZE_KiB_per_launch': write_kib,
       'correction': 'FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B)',
       'hbm_bytes_per_conv_launch': (2 * fetch_kib + write_kib) * 1024}
# ---- per launch: the k-th conv / stem dispatch of a step is the k-th conv row of bench.py's own launch table (--in-flight 1:
# dispatch order == launch order); algorithmic bytes from the same table
names = []
try:
    for line in open('%s/pmc_layers.txt' % out):
        p = line.split()
        if len(p) >= 5 and not line.startswith(('#', 'launch')) and float(p[-4]) > 0:
            names.append((p[0], float(p[-4])))
except OSError:
    pass
# dispatches per row: one, except the skinny-head pair where the patch kernel does not apply (split precision) - then each member is a
# launch of its own (launch_conv_group, kCfgPatch64)
disp = [1] * len(names)
steps_guess = [s for s in (7, 5) if names and len(seq['fetch']) % s == 0]
if names and steps_guess and len(seq['fetch']) // steps_guess[0] == len(names) + 1:
    for i, (nm, _) in enumerate(names):
        if nm.startswith('group[block4_objectness_score'):
            disp[i] = 2
L = sum(disp)
if L and len(seq['fetch']) % L == 0 and len(seq['fetch']) == len(seq['write']):
    steps = len(seq['fetch']) // L
    table, at = [], 0
    for i, (nm, gf) in enumerate(names):
        fk = sum(seq['fetch'][s * L + at + d][1] for s in range(steps) for d in range(disp[i])) / steps
        wk = sum(seq['write'][s * L + at + d][1] for s in range(steps) for d in range(disp[i])) / steps
        kern = seq['fetch'][at][0].split('<')[0].split('(')[0].replace('void ', '').split('::')[-1] + (' x%d' % disp[i] if disp[i] > 1 else '')
        at += disp[i]
        table.append((nm, kern, gf, 2 * fk * 1024 / 1e6, wk * 1024 / 1e6))
    with open('%s/traffic_layers_%s.txt' % (out, key), 'w') as f:
        f.write('# HBM traffic per launch (PMC, %d steps averaged; FETCH_SIZE x2, WRITE_SIZE), %s, commit %s\n' % (steps, res['command'], res['commit']))
        f.write('%-36s %-26s %9s %10s %10s %10s\n' % ('launch', 'kernel', 'GFLOP/img', 'read_MB', 'write_MB', 'total_MB'))
        for nm, kern, gf, rd, wr in table:
            f.write('%-36s %-26s %9.3f %10.1f %10.1f %10.1f\n' % (nm, kern, gf, rd, wr, rd + wr))
        f.write('%-36s %-26s %9.3f %10.1f %10.1f %10.1f\n' % ('TOTAL per step', '', sum(t[2] for t in table), sum(t[3] for t in table),
                                                               sum(t[4] for t in table), sum(t[3] + t[4] for t in table)))
    res['per_launch_table'] = 'traffic_layers_%s.txt' % key
    res['steps_sampled'] = steps
    res['launches_per_step_incl_stem'] = L
else:
    res['per_launch_table'] = None
    res['per_launch_note'] = 'dispatch sequence (%d fetch / %d write) is not a multiple of the %d conv rows of the launch table' % (len(seq['fetch']), len(seq['write']), L)
json.dump(res, open('%s/traffic_%s.json' % (out, key), 'w'), indent=1)
print(json.dumps(res))
PY
