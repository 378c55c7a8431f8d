#!/bin/bash
# residual loads of the transposed conv's epilogue batched per row block (one arrival wait per 4 rows instead of per row): A/B
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -2
for lib in prevres cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_$lib.so; fi
  echo "== $lib"; python3 tools/sweep_conv.py --cfgs=-1 --only b4_deconv,b5_deconv 2>&1 | tail -3
  python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 40 --warmup 10 --layers $O/res_${lib}_layers.txt > /dev/null 2>&1
done
paste <(awk 'NR>2{print $1, $3}' $O/res_prevres_layers.txt) <(awk 'NR>2{print $3}' $O/res_cur_layers.txt) | awk '{printf "%-40s before %7.1f after %7.1f\n", $1, $2, $3; a+=$2; b+=$3} END {print "sum", a, b}' | tail -12
for rep in 1 2 3; do for lib in prevres cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_$lib.so; fi
  python3 bench.py --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib default', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
