#!/bin/bash
# round 6: the whole GPU suite + smoke on the final tree, then the round's profile (tools/profile_round.sh)
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/final
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.txt 2>&1
tail -5 gpurun_out/final/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -2 gpurun_out/final/smoke.txt
bash tools/profile_round.sh > gpurun_out/final/profile_round.log 2>&1
tail -12 gpurun_out/final/profile_round.log
