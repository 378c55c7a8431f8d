set -u
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
L="conv3_2,conv4_2,b4_trio,b4_inc2,b4_quad,fc6_full,fc7_full,b4_cls,conv4_1,conv4_3,conv3_1,b5_trio,b5_inc2,b6_left_full"
python3 tools/sweep_conv.py --cfgs=-1,0,7 --only $L > $O/sweep_4w.txt 2>> $O/step2.err
RON_IGEMM256_V1=1 python3 tools/sweep_conv.py --cfgs=-1,0,7 --only $L > $O/sweep_8w.txt 2>> $O/step2.err
python3 tools/sweep_conv.py --cfgs=-1,0,7 --only $L > $O/sweep_4w_b.txt 2>> $O/step2.err
RON_IGEMM256_V1=1 python3 tools/sweep_conv.py --cfgs=-1,0,7 --only $L > $O/sweep_8w_b.txt 2>> $O/step2.err
echo "== 4 waves (asm)"; cat $O/sweep_4w.txt; echo "== 8 waves"; cat $O/sweep_8w.txt
echo "== 4 waves (asm) again"; cat $O/sweep_4w_b.txt; echo "== 8 waves again"; cat $O/sweep_8w_b.txt
python3 bench.py --no-cpu-baseline --no-parity-mode > $O/bench_4w_default.json 2>> $O/step2.err
RON_IGEMM256_V1=1 python3 bench.py --no-cpu-baseline --no-parity-mode > $O/bench_8w_default.json 2>> $O/step2.err
python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 > $O/bench_4w_if1.json 2>> $O/step2.err
RON_IGEMM256_V1=1 python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 > $O/bench_8w_if1.json 2>> $O/step2.err
for f in bench_4w_default bench_8w_default bench_4w_if1 bench_8w_if1; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1])
print('$f', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; done
tail -5 $O/step2.err
