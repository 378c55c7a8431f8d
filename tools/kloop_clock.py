#!/usr/bin/env python3
"""The shader clock the four-wave tiles' K loop runs at INSIDE the network, after 75 ms and after seconds of load.

Needs the diagnostic library (tools/build_stamps_variant.sh -> tools/experiments/libron_hip_stamps.so; the shipped kernels execute no
stamp): every 256 x 256 / 256 x 128 tile of a launch that does not split K writes {s_memtime, s_memrealtime} from in front of and
behind its assembly K loop into a ring of 65 536 records that nothing else reads.  In-kernel clock of a tile = delta s_memtime /
delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back, item 6); cycles per K step = delta s_memtime / steps.

  RON_HIP_LIB=$PWD/tools/experiments/libron_hip_stamps.so python3 tools/kloop_clock.py [--dtype bf16|f16x3] [--in-flight 2] [--seconds 3]

The bench.py workload (config 2), the same pipeline and loop.  Phases: (1) 5 + 20 steps from an idle GPU (what the driver's timed
region sees), ring read; (2) the loop for --seconds, ring read at the end = the last ~16 steps of it.  Per phase: images/s, median /
p10 / p90 clock over the tiles, median cycles per K step by tile width, clock by layer shape.
"""
import argparse
import json
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RING = 1 << 16


def summarize(ring, label, rate):
    n = int(ring[0])
    rec = ring[8:].reshape(RING, 8)
    rec = rec[:min(n, RING)]
    st, rt, steps, bn = (rec[:, 1] - rec[:, 0]).astype(np.float64), (rec[:, 3] - rec[:, 2]).astype(np.float64), rec[:, 4], rec[:, 5]
    ok = (rt > 100) & (st > 0) & (steps > 0)                 # a loop of >= 1 us; torn records (two launches, one slot) fail the range test below
    clk = np.where(ok, st / np.maximum(rt, 1) * 0.1, 0)
    ok &= (clk > 0.5) & (clk < 3.0)
    out = {'phase': label, 'images_per_s': rate, 'tiles_written': n, 'tiles_used': int(ok.sum())}
    if ok.sum() == 0:
        return out
    c = clk[ok]
    out['clock_ghz'] = {'median': float(np.median(c)), 'p10': float(np.percentile(c, 10)), 'p90': float(np.percentile(c, 90))}
    out['by_tile'] = {}
    for w in sorted(set(bn[ok].tolist())):
        m = ok & (bn == w)
        out['by_tile']['256x%d' % w] = {'tiles': int(m.sum()), 'clock_ghz': float(np.median(clk[m])),
                                        'cycles_per_k_step': float(np.median(st[m] / steps[m]))}
    out['by_shape'] = []
    shapes = sorted(set(zip(rec[ok, 6].tolist(), rec[ok, 7].tolist(), bn[ok].tolist())))
    for (M, KT, w) in shapes:
        m = ok & (rec[:, 6] == M) & (rec[:, 7] == KT) & (bn == w)
        out['by_shape'].append({'M': int(M), 'K_steps': int(KT), 'tile': '256x%d' % w, 'tiles': int(m.sum()), 'clock_ghz': float(np.median(clk[m])),
                                'cycles_per_k_step': float(np.median(st[m] / steps[m])), 'loop_us': float(np.median(rt[m]) * 0.01)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--in-flight', type=int, default=2)
    ap.add_argument('--seconds', type=float, default=3.0)
    ap.add_argument('--json', default='')
    a = ap.parse_args()
    import ctypes as C
    import torch
    from ron_tensorflow_amd import _lib, parallel
    from ron_tensorflow_amd.nets import nets_factory
    from ron_tensorflow_amd.pipeline import DetectPipeline
    from ron_tensorflow_amd.weights import synthetic_images, synthetic_weights
    lib = _lib.lib()
    try:
        lib.ron_debug_stamps
    except AttributeError:
        sys.exit('this library has no ron_debug_stamps: RON_HIP_LIB must point at tools/experiments/libron_hip_stamps.so')
    lib.ron_debug_stamps.argtypes = [C.c_void_p]
    lib.ron_debug_stamps.restype = C.c_int
    dev = torch.device('cuda', 0)
    cls = nets_factory.get_network('ron_320_vgg')
    net = cls(cls.default_params._replace(num_classes=21), variant='full', dtype=a.dtype, max_batch=a.batch, device=dev, fuse_pools=True)
    net.load_weights(synthetic_weights('full', seed=1))
    images = torch.from_numpy(synthetic_images(a.batch, seed=3)).to(dev)
    pipe = DetectPipeline(net, slots=a.in_flight, top_k=400)
    args = dict(objectness_thres=0.03, select_threshold=0.01, nms_threshold=0.45)
    ring = torch.zeros(8 + RING * 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    lib.ron_debug_stamps(C.c_void_p(ring.data_ptr()))
    phases = []
    res = parallel.bench_loop(pipe, images, 20, 5, a.in_flight, args, 400, device=dev)
    phases.append(summarize(ring.cpu().numpy(), 'burst: 5 + 20 steps from idle', a.batch * 20 / res['dt']))
    ring.zero_()
    torch.cuda.synchronize()
    ms = res['dt'] / 20 * 1e3
    steps = int(np.ceil(a.seconds * 1e3 / ms / 100)) * 100
    res = parallel.bench_loop(pipe, images, steps, 0, a.in_flight, args, 400, device=dev, window=100)
    s = summarize(ring.cpu().numpy(), 'sustained: last steps of %d (%.2f s)' % (steps, res['dt']), a.batch * steps / res['dt'])
    s["window_images_per_s"] = [round(100 * a.batch / (w * 1e-3), 1) for w in res['window_ms']]
    phases.append(s)
    lib.ron_debug_stamps(C.c_void_p(0))
    out = {'dtype': a.dtype, 'batch': a.batch, 'in_flight': a.in_flight, 'library': os.environ.get('RON_HIP_LIB', ''), 'phases': phases,
           'note': 'clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the assembly K loop of each four-wave tile (launches that do not '
                   'split K); the stamped library runs a few per cent slower than the shipped one (two stamps + one atomic per tile)'}
    for p in phases:
        print('# %s: %.0f images/s, %d tiles' % (p['phase'], p['images_per_s'], p.get('tiles_used', 0)))
        if 'clock_ghz' in p:
            print('#   clock GHz median %.3f (p10 %.3f, p90 %.3f)' % (p['clock_ghz']['median'], p['clock_ghz']['p10'], p['clock_ghz']['p90']))
            for k, v in p['by_tile'].items():
                print('#   %s: %d tiles, %.3f GHz, %.0f cycles per K step' % (k, v['tiles'], v['clock_ghz'], v['cycles_per_k_step']))
            for r in p['by_shape']:
                print('    M %7d  K steps %4d  %-8s tiles %6d  %.3f GHz  %6.0f cyc/step  loop %7.1f us' %
                      (r['M'], r['K_steps'], r['tile'], r['tiles'], r['clock_ghz'], r['cycles_per_k_step'], r['loop_us']))
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(out, f, indent=1)
    pipe.close()
    net.close()


if __name__ == '__main__':
    main()
