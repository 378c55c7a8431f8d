#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do
for lib in r04 cur; do
  if [ $lib = cur ]; then unset RON_HIP_LIB; else export RON_HIP_LIB=$PWD/tools/experiments/libron_hip_r04.so; fi
  python3 bench.py --no-cpu-baseline --no-parity-mode --batch 4 --in-flight 1 --steps 200 --warmup 30 --layers $O/ab_layers_b4_${lib}_$rep.txt > /dev/null 2>&1
done; done
paste <(awk 'NR>2{print $1, $3}' $O/ab_layers_b4_r04_1.txt) <(awk 'NR>2{print $3}' $O/ab_layers_b4_cur_1.txt) <(awk 'NR>2{print $3}' $O/ab_layers_b4_r04_2.txt) <(awk 'NR>2{print $3}' $O/ab_layers_b4_cur_2.txt) | awk '{printf "%-40s r04 %7.1f cur %7.1f | r04 %7.1f cur %7.1f\n", $1, $2, $3, $4, $5; a+=$2; b+=$3; c+=$4; d+=$5} END {print "sum", a, b, c, d}'
