#!/bin/bash
# round 6: the whole GPU suite (no -x), then the panel experiment
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_tests2
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
bash tools/experiments/r06_calls/r06_panels.sh
