mkdir -p gpurun_out/s3
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/s3/tests.log
python bench.py --batch 1 --in-flight 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity-mode > gpurun_out/s3/b1.json 2>gpurun_out/s3/b1.err
python bench.py --batch 1 --in-flight 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity-mode > gpurun_out/s3/b1_2.json 2>>gpurun_out/s3/b1.err
python bench.py --no-cpu-baseline --no-parity-mode > gpurun_out/s3/default.json 2>gpurun_out/s3/default.err
python bench.py --in-flight 1 --no-cpu-baseline --no-parity-mode > gpurun_out/s3/if1.json 2>gpurun_out/s3/if1.err
cat gpurun_out/s3/tests.log
for f in gpurun_out/s3/*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), round(d['ms_per_step'],4))"; done
