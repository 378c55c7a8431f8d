#!/bin/bash
# bench.py over small batches (one batch in flight): where the step is launch- / latency-bound rather than MFMA-bound
for b in ${BATCHES:-1 2 4 8 16 32}; do
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight ${FLIGHT:-1} --batch $b --steps ${STEPS:-100} --warmup 20 ${EXTRA} 2>/dev/null | tail -1 > /tmp/batch_$b.json
  python3 - "$b" <<'PY'
import json, sys
b = sys.argv[1]
d = json.loads(open('/tmp/batch_%s.json' % b).read())
print('batch', b, 'images/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'])
PY
done
