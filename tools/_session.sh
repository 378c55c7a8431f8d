O=gpurun_out/s6; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests.log
B="python bench.py --no-cpu-baseline --no-parity-mode"
RON_HIP_LIB=tools/experiments/libron_hip_prev.so $B --dtype f16x3 2>/dev/null | tail -1 > $O/f16x3_prev.json
$B --dtype f16x3 --layers $O/layers_f16x3.txt 2>/dev/null | tail -1 > $O/f16x3_new.json
RON_HIP_LIB=tools/experiments/libron_hip_prev.so $B --variant ssd512 --batch 16 2>/dev/null | tail -1 > $O/ssd_prev.json
$B --variant ssd512 --batch 16 2>/dev/null | tail -1 > $O/ssd_new.json
RON_HIP_LIB=tools/experiments/libron_hip_prev.so $B --variant reducedfc --dtype fp16 --batch 64 2>/dev/null | tail -1 > $O/cfg4_prev.json
$B --variant reducedfc --dtype fp16 --batch 64 2>/dev/null | tail -1 > $O/cfg4_new.json
RON_HIP_LIB=tools/experiments/libron_hip_prev.so $B --batch 1 --in-flight 1 --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/b1_prev.json
$B --batch 1 --in-flight 1 --steps 200 --warmup 20 --layers $O/layers_b1.txt 2>/dev/null | tail -1 > $O/b1_new.json
python tools/post_regimes.py > $O/post_regimes.txt 2>&1
cat $O/tests.log
for f in $O/*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['per_kernel_frac'],4))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
head -12 $O/layers_f16x3.txt; tail -12 $O/post_regimes.txt; tail -3 $O/layers_b1.txt
