#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_g8.py -x -q 2>&1 | tail -5
L="conv3_2,conv4_2,b4_trio,b4_inc2,fc6_full,fc7_full,b4_cls,conv4_1,conv3_1,b5_trio"
python3 tools/sweep_conv.py --dtype f16x3 --cfgs=-1,0 --only $L > $O/sweep_f16x3_4w.txt 2>> $O/step5.err
RON_IGEMM256_V1=1 python3 tools/sweep_conv.py --dtype f16x3 --cfgs=-1,0 --only $L > $O/sweep_f16x3_8w.txt 2>> $O/step5.err
echo "== 4w"; cat $O/sweep_f16x3_4w.txt; echo "== 8w"; cat $O/sweep_f16x3_8w.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --dtype f16x3 > $O/bench_f16x3_4w.json 2>> $O/step5.err
RON_IGEMM256_V1=1 python3 bench.py --no-cpu-baseline --no-parity-mode --dtype f16x3 > $O/bench_f16x3_8w.json 2>> $O/step5.err
for f in bench_f16x3_4w bench_f16x3_8w; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1])
print('$f', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; done
