#!/bin/bash
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
for b in 1 2 4; do for f in 1 2 3 4 6; do
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --batch $b --in-flight $f --steps 400 --warmup 40 2>/dev/null | tail -1 > /tmp/x.json
  python3 - $b $f <<'PY'
import json, sys
d = json.loads(open('/tmp/x.json').read())
print('batch', sys.argv[1], 'in_flight', sys.argv[2], 'images/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'])
PY
done; done
