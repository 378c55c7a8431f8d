#!/bin/bash
# persistent four-wave tiles, second form: the next tile's first LDS-DMA issued INSIDE the epilogue (behind the bias arrival, in front of the
# rows' stores), after the epilogue's per-row vmcnt waits were removed.  Parity forced on, then per-launch and whole-step A/B.
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
RON_IGEMM_PERSIST=1 timeout 1500 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_forward.py tests/test_gpu_benched_config.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
  RON_IGEMM_PERSIST=$v python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 40 --warmup 10 --layers $O/persist${v}_layers_$rep.txt 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persist $v in-flight 1', round(d['value'],1), round(d['ms_per_step'],3))"
  RON_IGEMM_PERSIST=$v python3 bench.py --no-cpu-baseline --no-parity-mode --steps 40 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persist $v default', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
paste <(awk 'NR>2{print $1, $3}' $O/persist0_layers_1.txt) <(awk 'NR>2{print $3}' $O/persist1_layers_1.txt) <(awk 'NR>2{print $3}' $O/persist0_layers_2.txt) <(awk 'NR>2{print $3}' $O/persist1_layers_2.txt) | awk '{printf "%-40s off %7.1f on %7.1f | off %7.1f on %7.1f\n", $1, $2, $3, $4, $5; a+=$2; b+=$3; c+=$4; d+=$5} END {print "sum", a, b, c, d}'
