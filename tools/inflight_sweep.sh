#!/bin/bash
# bench.py at 1..4 batches in flight (one line each): value, ms/step, conv roofline fraction
for f in ${FLIGHTS:-1 2 3 4}; do
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight $f --steps ${STEPS:-60} 2>/dev/null | tail -1 > /tmp/inflight_$f.json
  python3 - "$f" <<'PY'
import json, sys
f = sys.argv[1]
d = json.loads(open('/tmp/inflight_%s.json' % f).read())
print('in_flight', f, 'images/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])
PY
done
