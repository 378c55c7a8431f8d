#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for v in 0 1; do
  RON_IGEMM_PERSIST=$v python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 40 --warmup 10 --layers $O/persist${v}_layers_$rep.txt > /dev/null 2>&1
done; done
paste <(awk 'NR>2{print $1, $3}' $O/persist0_layers_1.txt) <(awk 'NR>2{print $3}' $O/persist1_layers_1.txt) <(awk 'NR>2{print $3}' $O/persist0_layers_2.txt) <(awk 'NR>2{print $3}' $O/persist1_layers_2.txt) | awk '{printf "%-40s off %7.1f on %7.1f | off %7.1f on %7.1f\n", $1, $2, $3, $4, $5; a+=$2; b+=$3; c+=$4; d+=$5} END {print "sum", a, b, c, d}'
