#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for f in 1 2 3 4; do python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight $f > $O/bench_if$f.json 2>/dev/null; python3 -c "
import json
d=json.loads([l for l in open('$O/bench_if$f.json') if l.startswith('{')][-1])
print('in-flight $f', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; done
