#!/bin/bash
# Board power, its cap and the reported shader clock while bench.py's sustained leg runs (rocm-smi sampled every 0.5 s) - context for
# "the K loop runs at the chip's power limit" (the in-kernel clock stamps are the measurement, this is the board's view).
# usage: tools/power_during_bench.sh <outfile> [bench.py args...]
set -u
export GPU_MAX_HW_QUEUES=8
OUT=$1; shift
python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 8 "$@" > $OUT.bench.json 2> /dev/null &
BP=$!
: > $OUT
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | tr '\n' ' ')" >> $OUT
  sleep 0.5
done
wait $BP
python3 - "$OUT" <<'PY'
import re, sys
rows = []
for l in open(sys.argv[1]):
    nums = re.findall(r'card0,([^ ]*)', l)
    if nums:
        rows.append(nums[0])
print('# samples', len(rows))
for r in rows[:3] + ['...'] + rows[-12:]:
    print(r)
PY
python3 - "$OUT.bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('bench value', round(d['value'], 1), 'sustained', round(d['sustained']['images_per_s'], 1), 'windows', d['sustained']['window_images_per_s']['all'])
PY
