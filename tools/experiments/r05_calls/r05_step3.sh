#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
echo "== 4-wave asm"; python3 tools/k_sweep.py | tee $O/k_sweep_4w.txt
echo "== 8-wave"; RON_IGEMM256_V1=1 python3 tools/k_sweep.py | tee $O/k_sweep_8w.txt
echo "== 4-wave asm N=2048"; python3 tools/k_sweep.py --n 2048 --rows 8192 | tee $O/k_sweep_4w_n2048.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --layers $O/layers_cfg2_if1_4w.txt > $O/bench_4w_if1_layers.json 2>> $O/step3.err
cat $O/layers_cfg2_if1_4w.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/step3_pytest_gpu.txt
