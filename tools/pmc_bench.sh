#!/bin/bash
# HBM traffic of the conv kernel from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (no tracing domains), FETCH_SIZE doubled on gfx950
# (it tallies 128-B requests at 64 B for wide coalesced streams), both counters are in KiB.
# usage: tools/pmc_bench.sh <outdir> [bench.py args...]      -> <outdir>/traffic_<variant>_<dtype>_bs<batch>.json
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1
python3 - "$OUT" "$@" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
args = sys.argv[2:]
def opt(name, default):
    return args[args.index(name) + 1] if name in args else default
key = '%s_%s_bs%s' % (opt('--variant', 'full'), opt('--dtype', 'bf16'), opt('--batch', '32'))
tot = {}
for which in ('fetch', 'write'):
    vals = []
    for f in glob.glob('%s/pmc_%s/*/*counter_collection.csv' % (out, which)):
        for r in csv.DictReader(open(f)):
            if any(k in r['Kernel_Name'] for k in ('conv_igemm', 'conv3x3_patch', 'conv3x3_c64')):      # every conv launch, grouped ones included
                vals.append(float(r['Counter_Value']))
    tot[which] = (sum(vals) / max(len(vals), 1), len(vals))
fetch_kib, n1 = tot['fetch']
write_kib, n2 = tot['write']
import os
res = {'command': 'bench.py --steps 5 --warmup 2 --no-cpu-baseline ' + ' '.join(args), 'commit': os.environ.get('RON_COMMIT'),
       'conv_launches_sampled': [n1, n2],
       'FETCH_SIZE_KiB_per_launch_raw': fetch_kib, 'WRITE_SIZE_KiB_per_launch': write_kib,
       'correction': 'FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B)',
       'hbm_bytes_per_conv_launch': (2 * fetch_kib + write_kib) * 1024}
json.dump(res, open('%s/traffic_%s.json' % (out, key), 'w'), indent=1)
print(json.dumps(res))
PY
