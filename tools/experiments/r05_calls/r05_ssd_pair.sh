#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ssd.py tests/test_gpu_benched_config.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do
  python3 bench.py --variant ssd512 --batch 16 --no-cpu-baseline --no-parity-mode --steps 60 --warmup 10 2>&1 | python3 -c "
import json,sys
L=[l for l in sys.stdin]
j=[l for l in L if l.startswith('{')]
print('ssd', (json.loads(j[-1])['value'], json.loads(j[-1])['ms_per_step']) if j else 'FAILED: '+''.join(L[-3:]))"
done
python3 bench.py --variant ssd512 --batch 16 --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 20 --warmup 5 --layers $O/ssd_pair_layers.txt > /dev/null 2>&1; tail -8 $O/ssd_pair_layers.txt
for b in 1 4; do python3 bench.py --variant ssd512 --batch $b --no-cpu-baseline --no-parity-mode --in-flight 1 --steps 100 --warmup 10 2>&1 | python3 -c "
import json,sys
j=[l for l in sys.stdin if l.startswith('{')]
print('ssd batch $b', json.loads(j[-1])['ms_per_step'])"; done
