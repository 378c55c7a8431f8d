#!/bin/bash
# Collects the round's judged artefacts on the MI355X box into gpurun_out/final/ (copy what matters into profiles/rNN/).
# usage: RON_COMMIT=<sha> tools/profile_round.sh
set -u
export TMPDIR=/tmp
# read by the HIP runtime when it initialises; under rocprofv3 that is before python starts (bench.py's setdefault comes too late there)
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/final
mkdir -p $O
python3 bench.py --layers $O/layers_cfg2_inflight2.txt > $O/bench_cfg2_default.json 2> $O/bench_cfg2_default.err
python3 bench.py --no-cpu-baseline --no-parity-mode --in-flight 1 --layers $O/layers_cfg2_inflight1.txt > $O/bench_cfg2_inflight1.json 2>> $O/err.txt
# the modes that meet north_star's 1e-4 clause: split precision (three f16 MFMAs per product) and exact-fp32 MFMA; both lines carry
# <dtype>_vs_fp32_oracle_agreement (cpu_baseline leg on)
python3 bench.py --dtype f16x3 --layers $O/layers_cfg2_f16x3.txt > $O/bench_cfg2_f16x3.json 2>> $O/err.txt
python3 bench.py --dtype f16x3 --no-cpu-baseline --no-parity-mode --in-flight 1 > $O/bench_cfg2_f16x3_inflight1.json 2>> $O/err.txt
python3 bench.py --dtype fp32 --steps 10 --warmup 3 --layers $O/layers_cfg2_fp32.txt > $O/bench_cfg2_fp32.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --variant reducedfc --dtype fp16 --batch 64 --layers $O/layers_cfg4.txt > $O/bench_cfg4.json 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --variant ssd512 --batch 16 --layers $O/layers_cfg5.txt > $O/bench_cfg5.json 2>> $O/err.txt
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --no-cpu-baseline --no-parity-mode --check-gather > $O/bench_cfg2_torchrun_1rank.json 2>> $O/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_cfg2_under_rocprof_inflight2.json 2>> $O/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_if1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 > $O/bench_cfg2_under_rocprof_inflight1.json 2>> $O/err.txt
for d in if1 if2; do f=$(ls $O/prof_$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$d.csv; done
# the kernel-level account of the DEFAULT mode (two batches in flight): wall, GPU busy, kernels resident, per launch the time it had the
# GPU to itself (tools/trace_overlap.py), from the trace of the same command
T=$(ls $O/prof_if2/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python3 tools/trace_overlap.py $T --layers $O/layers_cfg2_inflight2.txt --skip-steps 4 --json $O/trace_overlap_if2.json > $O/trace_overlap_if2.txt 2>&1
# the clock the K loop holds inside the network, after a burst and after seconds of load: s_memtime / s_memrealtime stamps of the
# diagnostic library (tools/build_stamps_variant.sh builds it; the shipped kernels execute no stamp)
if [ -f tools/experiments/libron_hip_stamps.so ]; then
  RON_HIP_LIB=$PWD/tools/experiments/libron_hip_stamps.so python3 tools/kloop_clock.py --json $O/kloop_clock_bf16_if2.json > $O/kloop_clock_bf16_if2.txt 2>> $O/err.txt
  RON_HIP_LIB=$PWD/tools/experiments/libron_hip_stamps.so python3 tools/kloop_clock.py --in-flight 1 --json $O/kloop_clock_bf16_if1.json > $O/kloop_clock_bf16_if1.txt 2>> $O/err.txt
  RON_HIP_LIB=$PWD/tools/experiments/libron_hip_stamps.so python3 tools/kloop_clock.py --dtype f16x3 --json $O/kloop_clock_f16x3_if2.json > $O/kloop_clock_f16x3_if2.txt 2>> $O/err.txt
fi
# PMC passes: ONE output directory per dtype (round 3's f16x3 files summed both dtypes' dispatches: same directory, globbed twice);
# the scripts also remove their counter directories before each pass
bash tools/pmc_bench.sh $O/pmc_bf16 --in-flight 1 > $O/pmc.log 2>&1
bash tools/pmc_mfma.sh $O/pmc_bf16 > $O/pmc_mfma.log 2>&1
bash tools/pmc_bench.sh $O/pmc_f16x3 --in-flight 1 --dtype f16x3 > $O/pmc_f16x3.log 2>&1
bash tools/pmc_mfma.sh $O/pmc_f16x3 --dtype f16x3 > $O/pmc_mfma_f16x3.log 2>&1
cp $O/pmc_bf16/traffic_*.json $O/pmc_bf16/traffic_layers_*.txt $O/pmc_bf16/mfma_busy_*.json $O/ 2>/dev/null
cp $O/pmc_f16x3/traffic_*.json $O/pmc_f16x3/traffic_layers_*.txt $O/pmc_f16x3/mfma_busy_*.json $O/ 2>/dev/null
python3 tools/post_regimes.py > $O/post_regimes.txt 2>> $O/err.txt
BATCHES="1 2 4 8 16 32" bash tools/batch_sweep.sh > $O/batch_sweep.txt 2>> $O/err.txt
# small batches: the two grouped plans of the heads side by side (contexts with max_batch <= 4 take the level plan by default)
echo "# --head-plan batch" >> $O/batch_sweep.txt
BATCHES="1 2 4 8" EXTRA="--head-plan batch" bash tools/batch_sweep.sh >> $O/batch_sweep.txt 2>> $O/err.txt
echo "# --head-plan level" >> $O/batch_sweep.txt
BATCHES="1 2 4 8" EXTRA="--head-plan level" bash tools/batch_sweep.sh >> $O/batch_sweep.txt 2>> $O/err.txt
python3 bench.py --no-cpu-baseline --no-parity-mode --batch 1 --in-flight 1 --steps 200 --warmup 20 --layers $O/layers_cfg2_batch1.txt > $O/bench_cfg2_batch1.json 2>> $O/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 --batch 1 > $O/bench_cfg2_batch1_under_rocprof.json 2>> $O/err.txt
f=$(ls $O/prof_b1/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_batch1.csv
rm -rf $O/prof_b1
# GEMM shapes beside hipBLASLt, and the same-box A/B against the previous round's library when one was left in tools/experiments
python3 tools/gemm_vs_vendor.py > $O/gemm_vs_vendor.txt 2>> $O/err.txt
if [ -n "${RON_PREV_LIB:-}" ] && [ -f "$RON_PREV_LIB" ]; then
  for rep in 1 2; do
    RON_HIP_LIB=$PWD/$RON_PREV_LIB python3 bench.py --no-cpu-baseline --no-parity-mode > $O/bench_cfg2_prevlib_$rep.json 2>> $O/err.txt
    python3 bench.py --no-cpu-baseline --no-parity-mode > $O/bench_cfg2_thislib_$rep.json 2>> $O/err.txt
  done
  echo "# previous round's library ($RON_PREV_LIB)" >> $O/batch_sweep.txt
  RON_HIP_LIB=$PWD/$RON_PREV_LIB BATCHES="1 2 4 8 16 32" bash tools/batch_sweep.sh >> $O/batch_sweep.txt 2>> $O/err.txt
fi
rm -rf $O/prof_if1 $O/prof_if2 $O/pmc_*/pmc_fetch $O/pmc_*/pmc_write $O/pmc_*/pmc_mfma
ls -la $O
tail -3 $O/pmc.log; tail -30 $O/pmc_mfma.log
for f in bench_cfg2_default bench_cfg2_inflight1 bench_cfg2_f16x3 bench_cfg2_f16x3_inflight1 bench_cfg2_fp32 bench_cfg4 bench_cfg5 bench_cfg2_torchrun_1rank; do python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
    r=d['roofline']
    print(sys.argv[1], round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'per_kernel', round(r.get('per_kernel_frac',0),3))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
