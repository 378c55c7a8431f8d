#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_power
rocm-smi --showpower --showclocks --showuse --csv 2>&1 | head -5
rocm-smi --showmaxpower 2>&1 | tail -4
bash tools/power_during_bench.sh gpurun_out/r06_power/power_bf16_if2.txt
bash tools/power_during_bench.sh gpurun_out/r06_power/power_bf16_if1.txt --in-flight 1
bash tools/power_during_bench.sh gpurun_out/r06_power/power_f16x3_if2.txt --dtype f16x3
