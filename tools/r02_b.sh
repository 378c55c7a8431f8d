#!/bin/bash
# Round-2 second GPU pass: the rebuilt conv kernels (shared halos, blocked weights, flat / pixel-tile halo-patch kernel).
set -u
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
timeout 900 python3 tools/sweep_conv.py --cfgs=-1,0,1,2,3,4,5,6 > $O/sweep_all.txt 2>&1
cat $O/sweep_all.txt
python3 bench.py --no-cpu-baseline --layers $O/layers_default.txt > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --no-cpu-baseline --in-flight 1 --layers $O/layers_if1.txt > $O/bench_if1.json 2> $O/bench_if1.err
for f in bench_default bench_if1; do python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
    r=d['roofline']
    print(sys.argv[1], round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'per_kernel', round(r.get('per_kernel_frac',0),3))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $O/layers_if1.txt
