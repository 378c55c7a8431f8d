#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/step4_pytest_gpu.txt
python3 bench.py --no-cpu-baseline --no-parity-mode > $O/bench_epi_default.json 2>> $O/step4.err
python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --layers $O/layers_cfg2_if1_epi.txt > $O/bench_epi_if1.json 2>> $O/step4.err
python3 bench.py --no-cpu-baseline --no-parity-mode --dtype f16x3 --layers $O/layers_cfg2_f16x3_epi.txt > $O/bench_epi_f16x3.json 2>> $O/step4.err
python3 bench.py --no-cpu-baseline --no-parity-mode --variant ssd512 --batch 16 > $O/bench_epi_cfg5.json 2>> $O/step4.err
python3 bench.py --no-cpu-baseline --no-parity-mode --variant reducedfc --dtype fp16 --batch 64 > $O/bench_epi_cfg4.json 2>> $O/step4.err
for f in bench_epi_default bench_epi_if1 bench_epi_f16x3 bench_epi_cfg5 bench_epi_cfg4; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1])
print('$f', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; done
cat $O/layers_cfg2_if1_epi.txt
cat $O/layers_cfg2_f16x3_epi.txt
