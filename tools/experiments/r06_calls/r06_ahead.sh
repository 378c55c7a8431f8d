#!/bin/bash
# round 6: does bounding the host's run-ahead remove the window-to-window dips of the sustained leg?
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_ahead
mkdir -p $O
for A in 0 4 8 16 32 0; do
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 4 --max-ahead $A > $O/bench_A$A.json 2>> $O/err.txt
  python3 - "$O/bench_A$A.json" $A <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l); s=d['sustained']; w=s['window_images_per_s']
print('max_ahead', sys.argv[2], 'timed', round(d['value'],1), 'sustained', round(s['images_per_s'],1), 'windows', w['all'])
PY
done
