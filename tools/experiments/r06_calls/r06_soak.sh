#!/bin/bash
# round 6: determinism soak of the final library, the in-flight sweep with host flow control
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_soak
mkdir -p $O
timeout 1200 python3 tools/soak_determinism.py 60 > $O/soak_determinism.txt 2>&1
tail -14 $O/soak_determinism.txt
bash tools/inflight_sweep.sh > $O/inflight_sweep.txt 2>&1
cat $O/inflight_sweep.txt
