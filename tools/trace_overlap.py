#!/usr/bin/env python3
"""Kernel-level account of a bench.py run with SEVERAL batches in flight, from `rocprofv3 --kernel-trace` (…_kernel_trace.csv).

With two execution slots the launches of two steps overlap, so a launch's duration is no longer its kernel's alone and the durations
of a step add up to more than the step.  This tool says what the trace does show:

  * wall time of the steady-state region, the union of all kernel intervals (GPU busy), the sum of kernel durations, sum / wall
    (= average number of kernels resident), idle time;
  * per launch position of a step (named from a bench.py --layers file when given): average duration here, and its EXPOSED time =
    the part of its interval in which it is the only kernel on the GPU - the share of the step that launch alone is answerable for.
    A small launch that is never alone costs nothing but the CUs it occupies; one that is always alone is serial time.

usage: tools/trace_overlap.py <kernel_trace.csv> [--layers layers.txt] [--skip-steps 3] [--json out.json]
Steps are cut per stream at `topk_nms_kernel` (the last kernel of ron_detect); split-K finalize kernels and the post-processing
kernels are attached to the launch in front of them.
"""
import argparse
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r'(?:ron::detail::|\(anonymous namespace\)::)+(\w+)', name)
    if not m:
        return name[:40]
    dims = re.search(r'Traits\w+, (\d+), (\d+), \d+, \d+, \d+, \d+>', name)
    return m.group(1).replace('_kernel', '') + ('<%sx%s>' % dims.groups() if dims else '')


def read_trace(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Kind', 'KERNEL_DISPATCH') != 'KERNEL_DISPATCH':
                continue
            rows.append(dict(q=(r.get('Queue_Id'), r.get('Stream_Id')), name=r['Kernel_Name'], s=int(r['Start_Timestamp']), e=int(r['End_Timestamp']),
                             grid=int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)))
    rows.sort(key=lambda r: r['s'])
    return rows


def layer_names(path):
    names = []
    for line in open(path):
        if line.startswith('#') or line.startswith('launch'):
            continue
        p = line.split()
        if len(p) >= 5:
            names.append((p[0], float(p[2])))
    return names


ATTACHED = ('splitk_finalize', 'topk_partial', 'topk_nms')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--layers', default='')
    ap.add_argument('--skip-steps', type=int, default=3, help='steps per stream dropped at the start (warm-up) ')
    ap.add_argument('--json', default='')
    a = ap.parse_args()
    rows = read_trace(a.trace)
    by_q = collections.defaultdict(list)
    for r in rows:
        by_q[r['q']].append(r)
    # slot streams: the ones that run ron_detect's last kernel
    slots = {q: v for q, v in by_q.items() if any('topk_nms_kernel' in r['name'] for r in v)}
    if not slots:
        sys.exit('no topk_nms_kernel in the trace: not a ron_detect run')
    steps = []                                   # (queue, [rows of one step])
    for q, v in slots.items():
        cur, mine = [], []
        for r in v:
            cur.append(r)
            if 'topk_nms_kernel' in r['name']:
                mine.append(cur)
                cur = []
        steps += [(q, s) for s in mine[a.skip_steps:]]
    length = collections.Counter(len(s) for _, s in steps).most_common(1)[0][0]
    steps = [(q, s) for q, s in steps if len(s) == length]
    if not steps:
        sys.exit('no complete steps')
    t0 = min(s[0]['s'] for _, s in steps)
    t1 = max(s[-1]['e'] for _, s in steps)
    inside = [r for r in rows if r['e'] > t0 and r['s'] < t1]
    # sweep: number of kernels resident between consecutive event times
    ev = sorted({t0, t1} | {min(max(r['s'], t0), t1) for r in inside} | {min(max(r['e'], t0), t1) for r in inside})
    import bisect
    cnt = [0] * (len(ev) - 1)
    for r in inside:
        i0, i1 = bisect.bisect_left(ev, max(r['s'], t0)), bisect.bisect_left(ev, min(r['e'], t1))
        for i in range(i0, i1):
            cnt[i] += 1
    seg = [ev[i + 1] - ev[i] for i in range(len(cnt))]
    wall = t1 - t0
    busy = sum(d for d, c in zip(seg, cnt) if c > 0)
    total = sum(d * c for d, c in zip(seg, cnt))
    hist = collections.Counter()
    for d, c in zip(seg, cnt):
        hist[min(c, 4)] += d

    def exposed(r):
        i0, i1 = bisect.bisect_left(ev, max(r['s'], t0)), bisect.bisect_left(ev, min(r['e'], t1))
        return sum(seg[i] for i in range(i0, i1) if cnt[i] == 1)

    names = layer_names(a.layers) if a.layers else []
    pos = []                                     # per position of a step
    for i in range(length):
        rs = [s[i] for _, s in steps]
        pos.append(dict(kernel=short(rs[0]['name']), grid=rs[0]['grid'], n=len(rs), dur_us=sum(r['e'] - r['s'] for r in rs) / len(rs) * 1e-3,
                        exposed_us=sum(exposed(r) for r in rs) / len(rs) * 1e-3))
    # group positions into launches: a primary kernel + the attached kernels behind it
    launches = []
    for p in pos:
        if launches and any(k in p['kernel'] for k in ATTACHED):
            launches[-1]['parts'].append(p)
        else:
            launches.append(dict(parts=[p]))
    if names and len(names) == len(launches):
        for l, (nm, solo) in zip(launches, names):
            l['name'], l['solo_us'] = nm, solo
    n_steps = len(steps)
    step_ms = wall / n_steps * 1e-6
    out = dict(trace=a.trace, steps=n_steps, streams=len(slots), kernels_per_step=length, wall_ms=wall * 1e-6, ms_per_step=step_ms,
               gpu_busy_ms=busy * 1e-6, idle_ms=(wall - busy) * 1e-6, sum_kernel_ms=total * 1e-6, overlap_factor=total / wall,
               resident_histogram={('%d' % k if k < 4 else '4+'): v / wall for k, v in sorted(hist.items())}, launches=[])
    print('# %s' % a.trace)
    print('# %d steps on %d streams, %d kernels per step; region %.3f ms = %.3f ms per step' % (n_steps, len(slots), length, wall * 1e-6, step_ms))
    print('# GPU busy %.3f ms (idle %.1f %%), sum of kernel durations %.3f ms -> %.2f kernels resident on average' %
          (busy * 1e-6, 100 * (wall - busy) / wall, total * 1e-6, total / wall))
    print('# share of the region with k kernels resident: ' + ', '.join('%s: %.1f %%' % (('%d' % k if k < 4 else '4+'), 100 * v / wall) for k, v in sorted(hist.items())))
    print('%-34s %-34s %9s %9s %9s %8s' % ('launch', 'kernel', 'dur_us', 'alone_us', 'solo_us', 'alone_%'))
    tot_exp = 0.0
    for l in launches:
        d = sum(p['dur_us'] for p in l['parts'])
        e = sum(p['exposed_us'] for p in l['parts'])
        tot_exp += e
        kern = '+'.join(p['kernel'] for p in l['parts'])
        row = dict(name=l.get('name', ''), kernel=kern, dur_us=d, alone_us=e, solo_us=l.get('solo_us'), alone_share_of_step=e / (step_ms * 1e3))
        out['launches'].append(row)
        print('%-34s %-34s %9.1f %9.1f %9s %8.2f' % (row['name'][:34], kern[:34], d, e, '%.1f' % l['solo_us'] if 'solo_us' in l else '-',
                                                    100 * row['alone_share_of_step']))
    print('# alone time per step %.1f us of %.1f (%.1f %%): the part of a step during which ONE kernel had the GPU to itself' %
          (tot_exp, step_ms * 1e3, 100 * tot_exp / (step_ms * 1e3)))
    out['alone_us_per_step'] = tot_exp
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
