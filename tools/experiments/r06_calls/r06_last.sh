#!/bin/bash
# round 6, last call: the whole GPU suite, smoke and the driver's own command on the final tree
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_last
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -1
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_last/bench_driver_cmd.json') if l.startswith('{')][-1])
print('value', round(d['value'],1), 'sustained', round(d['sustained']['images_per_s'],1), 'parity', round(d['parity_mode']['images_per_s'],1), 'parity sustained', round(d['parity_mode']['sustained']['images_per_s'],1))
print('roofline frac', round(d['roofline']['frac'],3), 'per_kernel', round(d['roofline']['per_kernel_frac'],3), 'mfma_busy src', d['roofline']['mfma_busy_source']['file'], 'traffic src', d['roofline']['traffic_source']['file'], 'clock src', d['roofline']['k_loop_clock_source']['file'])
PY
