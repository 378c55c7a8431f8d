#!/bin/bash
# run-to-run spread of the default line on ONE box (what an A/B difference has to be read against)
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 3 2>/dev/null | tail -1 > /tmp/n.json
  python3 - <<'PY'
import json
d = json.loads(open('/tmp/n.json').read())
print('timed %.1f  sustained %.1f  windows min %.0f max %.0f' % (d['value'], d['sustained']['images_per_s'], d['sustained']['window_images_per_s']['min'], d['sustained']['window_images_per_s']['max']))
PY
done
