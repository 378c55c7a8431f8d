#!/usr/bin/env python3
"""Prints the figures DESIGN.md section 8 / README / profiles/README quote, from the files tools/profile_round.sh leaves in a directory
(default gpurun_out/final, or profiles/rNN with the final_ prefix).   python3 tools/summarize_final.py [dir] [prefix]"""
import csv
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/final'
pre = sys.argv[2] if len(sys.argv) > 2 else ''


def line(name):
    p = os.path.join(d, pre + name)
    return json.loads([x for x in open(p) if x.startswith('{')][-1])


for name in ('bench_cfg2_default', 'bench_cfg2_inflight1', 'bench_cfg2_f16x3', 'bench_cfg2_f16x3_inflight1', 'bench_cfg2_fp32', 'bench_cfg4',
             'bench_cfg5', 'bench_cfg2_torchrun_1rank', 'bench_cfg2_batch1', 'bench_cfg2_prevlib_1', 'bench_cfg2_thislib_1', 'bench_cfg2_prevlib_2',
             'bench_cfg2_thislib_2', 'bench_cfg2_under_rocprof_inflight1'):
    try:
        j = line(name + '.json')
    except Exception as e:
        print(name, 'missing', e)
        continue
    r = j['roofline']
    print('%-36s %8.1f images/s  %7.3f ms  conv %6.0f TF  frac %.3f  per-kernel %.3f  launch %.1f us' % (
        name, j['value'], j['ms_per_step'], r['achieved'], r['frac'], r.get('per_kernel_frac', 0), r.get('avg_launch_us', 0)))
    if 'sustained' in j:
        su = j['sustained']
        w = su['window_images_per_s']
        print('    sustained %.1f images/s over %.2f s (%d steps): windows min %.0f median %.0f max %.0f, last / first %.3f, burst / sustained %.4f' % (
            su['images_per_s'], su['seconds'], su['steps'], w['min'], w['median'], w['max'], su['last_over_first_window'], su['burst_over_sustained']))
    if 'parity_mode' in j:
        pm = j['parity_mode']
        print('    parity_mode %.1f images/s %.2f ms, agreement %s' % (pm['images_per_s'], pm['ms_per_step'], pm.get('agreement')))
        if 'sustained' in pm:
            su = pm['sustained']
            print('    parity_mode sustained %.1f images/s over %.2f s, burst / sustained %.4f' % (su['images_per_s'], su['seconds'], su['burst_over_sustained']))
    if 'cpu_baseline' in j:
        cb = j['cpu_baseline']
        print('    cpu %.2f images/s (batch 1: %.2f)' % (cb['value'], cb['batch_1']['end_to_end_images_per_s']))
    for k in j:
        if k.endswith('oracle_agreement'):
            print('    %s reproduced %.4f within %.4f' % (k, j[k]['reproduced'], j[k]['within_1e-4_of_reproduced']))
rows = list(csv.DictReader(open(os.path.join(d, pre + 'kernel_stats_if1.csv'))))
conv = [r for r in rows if 'conv' in r['Name'] and 'splitk' not in r['Name']]
tot = sum(float(r['TotalDurationNs']) for r in conv)
n = sum(int(r['Calls']) for r in conv)
print('rocprof, one in flight: %d conv launches, %.2f ms, %.1f us per launch' % (n, tot / 1e6, tot / n / 1e3))
for r in rows:
    if any(k in r['Name'] for k in ('select_kernel', 'topk_nms', 'topk_partial', 'splitk_finalize', 'stem2')):
        print('    %-60s %4s calls  %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
for f in ('traffic_full_bf16_bs32', 'traffic_full_f16x3_bs32'):
    t = json.load(open(os.path.join(d, f + '.json')))
    print('%s: %.1f MB per conv launch (commit %s)' % (f, t['hbm_bytes_per_conv_launch'] / 1e6, t['commit']))
for f in ('mfma_busy_full_bf16_bs32', 'mfma_busy_full_f16x3_bs32'):
    t = json.load(open(os.path.join(d, f + '.json')))
    w = t['whole_run']
    print('%s: whole run %.3f at %.2f GHz (x clock / nominal %.3f)' % (f, w['mfma_busy_fraction'], w['clock_ghz'], w['mfma_busy_x_clock_over_nominal']))
    for k, v in t['per_kernel'].items():
        if 'conv' in k:
            print('    %-48s %.3f at %.2f GHz' % (k, v['mfma_busy_fraction'], v['clock_ghz']))
for f in ('batch_sweep.txt', 'post_regimes.txt', 'gemm_vs_vendor.txt'):
    print('--', f)
    print(open(os.path.join(d, pre + f)).read().rstrip()[:2400])
print('-- traffic per step')
print(open(os.path.join(d, 'traffic_layers_full_bf16_bs32.txt')).read().rstrip().split('\n')[-1])
