#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
python3 bench.py --no-cpu-baseline --no-parity-mode > $O/b_n128_default.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --layers $O/layers_n128_if1.txt > $O/b_n128_if1.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-parity-mode --dtype f16x3 > $O/b_n128_f16x3.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-parity-mode --variant ssd512 --batch 16 > $O/b_n128_cfg5.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-parity-mode --variant reducedfc --dtype fp16 --batch 64 > $O/b_n128_cfg4.json 2>/dev/null
for f in b_n128_default b_n128_if1 b_n128_f16x3 b_n128_cfg5 b_n128_cfg4; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1])
print('$f', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; done
grep -E "conv2_|conv1_" $O/layers_n128_if1.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
