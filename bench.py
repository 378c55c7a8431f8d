#!/usr/bin/env python3
"""bench.py -- images/sec of RON-VGG16-320 inference on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic pre-whitened 320x320x3 images that are
already resident in HBM: conv stack (ron_net, full VGG-16 fc6/fc7, bf16 MFMA) -> softmax + objectness gate ->
decode -> select -> top-k -> class-aware NMS (np_methods semantics), i.e. one ron_detect() call; with N > 1
every rank processes its own shard of images (weak scaling) and the fixed-size detection records are
all-gathered over RCCL (the only exchange the path has).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task description) with `roofline` (dominant kernel = the
implicit-GEMM conv kernel, MFMA-bound; achieved = algorithmic conv FLOPs per launch / HIP-event duration of
those launches inside the timed region) and, at N = 1, `cpu_baseline` (the oracle port timed on the host cores) and
`parity_mode` (the same workload in dtype f16x3, the arithmetic that meets the 1e-4 clause: its own timed steps after the
headline's timed region, with its agreement against the fp32 oracle).
"""
import argparse
import json
import os
import sys
import time

# Streams in use per process: two execution slots + the consumer + RCCL's own stream + the default stream.  The HIP runtime
# maps streams onto 4 hardware queues by default, so two of them share one and serialise behind each other (measured: the
# single-rank RCCL path 1.8 % behind the plain one with 4 queues, level with it with 8).  Must be set before HIP initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F16_TFLOPS = 2500.0
PEAK_F32_TFLOPS = 157.3
PEAK_F16X3_TFLOPS = 2500.0 / 3   # split precision: three f16 MFMAs per algorithmic product
PROFILED_STEPS = 3
PARITY_WARMUP = 10
SUSTAINED_WINDOW = 100         # steps per window of the sustained leg


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step (BASELINE config 2: 32)')
    ap.add_argument('--variant', default='full', choices=['full', 'reducedfc', 'ssd512'],
                    help="full = BASELINE configs 2/3 (default); reducedfc = config 4; ssd512 = config 5")
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16', 'fp32', 'f16x3'],
                    help="bf16 = BASELINE config 2 (default); f16x3 = split precision: detections within 1e-4 of the fp32 CPU reference "
                         "on the f16 matrix cores (three MFMAs per product); fp32 = exact-fp32 MFMA parity mode")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-mode', action='store_true',
                    help='skip the second, untimed-by-the-headline leg that runs the same workload in the arithmetic that meets '
                         'north_star\'s 1e-4 clause (dtype f16x3) and reports it as "parity_mode"')
    ap.add_argument('--parity-steps', type=int, default=20)
    ap.add_argument('--sustained-seconds', type=float, default=3.0,
                    help='after the timed region: the same loop for at least this long without a host wait inside it ("sustained" in the '
                         'JSON line: overall rate and the rate of every window of %d steps); the same for parity_mode.  0 = skip' % SUSTAINED_WINDOW)
    ap.add_argument('--cpu-images', type=int, default=6, help='images run one by one (batch 1) in the bounded CPU-baseline sample; '
                    'one batch of up to --batch images follows')
    ap.add_argument('--multi-stream', action='store_true',
                    help='run the block7/6/5 head branches on side streams (RON_CFG_MULTI_STREAM, round 1; it excludes the grouped '
                         'launches that replaced it and is slower than the default now: 5.2 k vs 6.9 k images/s with two batches in '
                         'flight).  Kept for A/B runs and the bitwise-equivalence test')
    ap.add_argument('--head-plan', default=None, choices=['level', 'batch'],
                    help='grouped launch plan of the RON heads (default: by batch, RON_CFG_LEVEL_GROUPS / RON_CFG_BATCH_GROUPS)')
    ap.add_argument('--in-flight', type=int, default=2,
                    help='batches in flight per GPU (execution slots over one set of weights, one stream each; '
                         'ron_tensorflow_amd/pipeline.py).  1 = strictly one launch at a time: per-launch durations are then '
                         "each kernel's alone, which is what profiles/*/kernel_stats are taken with")
    ap.add_argument('--max-queued', type=int, default=8,
                    help='host flow control of the pipeline (DetectPipeline.max_queued): submit() waits while that many submitted batches '
                         'have not finished on the GPU; 0 = unbounded (the host enqueues as fast as it can: 1.3 %% slower over the timed '
                         'region, window-to-window dips of 10 %% in the sustained leg)')
    ap.add_argument('--layers', default='', help='write the per-launch timing table to this file')
    ap.add_argument('--check-gather', action='store_true', default=True,
                    help='under torch.distributed.run (default on): after the timed region every rank compares the all-gathered '
                         'records with its own local ones and checks every rank\'s counts; the JSON line gets "gather_check"')
    ap.add_argument('--no-check-gather', dest='check_gather', action='store_false')
    return ap.parse_args()


def physical_cores():
    """Physical cores this process may run on (SMT siblings counted once), from /proc/cpuinfo; falls back to the affinity count."""
    allowed = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else set(range(os.cpu_count() or 1))
    try:
        cores, cpu, phys = set(), None, 0
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('processor'):
                    cpu = int(line.split(':')[1])
                elif line.startswith('physical id'):
                    phys = int(line.split(':')[1])
                elif line.startswith('core id') and cpu in allowed:
                    cores.add((phys, int(line.split(':')[1])))
        if cores:
            return len(cores), len(allowed)
    except (OSError, ValueError):
        pass
    return len(allowed), len(allowed)


def cpu_baseline(variant, weights, images, n_batch1, big_batch):
    """The oracle port on the host cores: fp32 conv stack on the torch-CPU operators with
    torch.set_num_threads(physical cores) + numpy np_methods post-processing, at batch 1 (`n_batch1` images one by one)
    and at one batch of `big_batch`, conv stack / post-processing / end to end timed separately.  `images`: the bench
    batch itself (host copy), so the detections double as the fp32 reference of the agreement figure.
    Returns (json dict, per-image fp32-oracle detections of the batch-1 images)."""
    import torch
    from oracle import anchors as oanchors
    from oracle import np_post
    from oracle import ron_forward as orf
    from oracle import ssd_forward as osf
    ssd = variant == 'ssd512'
    anchors = osf.anchors_all_layers() if ssd else oanchors.anchors_all_layers()
    cores, logical = physical_cores()
    torch.set_num_threads(cores)

    def forward(batch):
        if ssd:
            pred, loc, _, _ = osf.ssd_forward(batch, weights, backend='torch')
            return pred, loc, None
        pred, _, objp, _, loc, _ = orf.ron_forward(batch, weights, variant, backend='torch')
        return pred, loc, objp

    def run(batch):
        t0 = time.perf_counter()
        pred, loc, objp = forward(batch)
        t1 = time.perf_counter()
        det = np_post.detect_from_predictions(pred, loc, anchors, objness_pred=objp)
        return det, t1 - t0, time.perf_counter() - t1

    run(images[:1])                              # warm-up (thread pool, page faults)
    dets, conv1, post1 = [], 0.0, 0.0
    for i in range(n_batch1):
        d, tc, tp = run(images[i:i + 1])
        dets.append(d[0])
        conv1 += tc
        post1 += tp
    # one large batch, bounded to ~30 s of host time (a batched conv stack runs >= 2.5x the batch-1 rate per image on a
    # many-core host: 3.7x measured on 128 cores)
    per_image = (conv1 + post1) / max(n_batch1, 1)
    nb = int(max(1, min(big_batch, len(images), 30.0 * 2.5 // max(per_image, 1e-3))))
    _, convb, postb = run(images[:nb])

    def rates(n, tc, tp):
        return {'images': n, 'conv_stack_images_per_s': n / tc, 'post_images_per_s': n / tp, 'end_to_end_images_per_s': n / (tc + tp),
                'seconds': tc + tp}

    b1, bb = rates(n_batch1, conv1, post1), rates(nb, convb, postb)
    best = max(b1['end_to_end_images_per_s'], bb['end_to_end_images_per_s'])
    out = {'value': best, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
           'threads': cores, 'logical_cpus': logical, 'batch_1': b1, 'batch_%d' % nb: bb,
           'sample': '%s, fp32: conv stack on torch-CPU operators with torch.set_num_threads(%d physical cores) + numpy np_methods '
                     'post-processing; %d images at batch 1 (%.1f s) and one batch of %d (%.1f s); value = the better end-to-end rate'
                     % (variant, cores, n_batch1, conv1 + post1, nb, convb + postb)}
    return out, dets


def load_traffic(args):
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes of this same command (tools/pmc_bench.sh ->
    profiles/<round>/traffic_*.json).  NOT measured in this run: returned with its provenance (file, and the commit /
    command the file records), or (None, None) when no measurement exists for this configuration."""
    import glob
    key = '%s_%s_bs%d' % (args.variant, args.dtype, args.batch)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'traffic_%s.json' % key)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    src = {'file': os.path.relpath(files[-1], ROOT), 'commit': t.get('commit'), 'command': t.get('command'),
           'note': 'committed PMC measurement of an earlier run of this command (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes); '
                   'not collected in this run'}
    return t.get('hbm_bytes_per_conv_launch'), src


def load_mfma_busy(args):
    """Matrix-pipe occupancy of this command by counter (tools/pmc_mfma.sh -> profiles/<round>/mfma_busy_*.json: SQ_VALU_MFMA_BUSY_CYCLES
    over elapsed cycles, the clock the dispatches ran at, and busy x clock / 2.4 GHz = fraction of the nominal matrix peak).  Like
    `traffic`: a committed measurement of an earlier run of the same command, returned with its provenance, or (None, None)."""
    import glob
    key = '%s_%s_bs%d' % (args.variant, args.dtype, args.batch)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'mfma_busy_%s.json' % key)) +
                   glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'final_mfma_busy_%s.json' % key)),
                   key=lambda f: (os.path.basename(os.path.dirname(f)), os.path.basename(f)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    w = t.get('whole_run', {})
    # (round 6: the clock those files derive from GRBM_GUI_ACTIVE / dispatch duration, and busy x clock / 2.4 GHz, are no longer carried:
    # on dispatches of 60-300 us that quotient reads high - MI355X_MICROARCH.md, DVFS give-back - and the in-kernel stamps of
    # `roofline.k_loop_clock` are the measurement of the clock)
    val = {'whole_run': w.get('mfma_busy_fraction'),
           'per_kernel': {k: v.get('mfma_busy_fraction') for k, v in t.get('per_kernel', {}).items() if 'conv' in k or 'stem' in k}}
    src = {'file': os.path.relpath(files[-1], ROOT), 'commit': t.get('commit'), 'command': t.get('command'),
           'note': 'committed rocprofv3 --pmc measurement of an earlier run of this command, one batch in flight (SQ_VALU_MFMA_BUSY_CYCLES '
                   '/ (1024 SIMDs x GRBM_GUI_ACTIVE / 8)); not collected in this run'}
    return val, src


def sustained_leg(pipe, images, ms_per_step_estimate, seconds, in_flight, detect_args, top_k, batch, burst_images_per_s, **loop_args):
    """The timed loop again, for >= `seconds` (a whole number of windows of SUSTAINED_WINDOW steps, at least four), no warm-up of its
    own and no host wait inside: the clock a serving loop holds, which a 20-step timed region (75 ms) cannot show (the chip lowers its
    clock under seconds of load: MI355X_MICROARCH.md, DVFS give-back).  Rates per window from timing events on the consuming stream;
    the first window's steps lead up to the first event (parallel.bench_loop), so N windows of steps give N - 1 window rates."""
    from ron_tensorflow_amd import parallel
    w = SUSTAINED_WINDOW
    steps = max(4, int(np.ceil(seconds * 1e3 / max(ms_per_step_estimate, 1e-3) / w))) * w
    res = parallel.bench_loop(pipe, images, steps, 0, in_flight, detect_args, top_k, check_gather=False, window=w, measure_gather=False,
                              **loop_args)
    dt, win = res['dt'], res['window_ms']
    rates = [w * batch / (ms * 1e-3) for ms in win]
    overall = batch * steps / dt
    return {'seconds': dt, 'steps': steps, 'images_per_s': overall, 'ms_per_step': dt / steps * 1e3, 'window_steps': w,
            'window_images_per_s': {'first': rates[0], 'last': rates[-1], 'min': min(rates), 'median': float(np.median(rates)),
                                    'max': max(rates), 'all': [round(r, 1) for r in rates]},
            'last_over_first_window': rates[-1] / rates[0],
            'burst_over_sustained': burst_images_per_s / overall,
            'note': 'same pipeline, images and loop as the timed region, started right after it; whole job (rank 0\'s marks); burst = the '
                    'timed region of this line'}


def load_kloop_clock(args, in_flight):
    """The shader clock inside the assembly K loop of the four-wave tiles while this command runs: s_memtime / s_memrealtime stamps of a
    DIAGNOSTIC build of the library (tools/build_stamps_variant.sh, tools/kloop_clock.py -> profiles/<round>/kloop_clock_<dtype>_if<F>.json),
    after a burst from idle (what the driver's 5 + 20 steps see) and after seconds of load.  A committed measurement like `traffic`
    (the shipped kernels execute no stamp): returned with its provenance, or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'kloop_clock_%s_if%d.json' % (args.dtype, in_flight))))
    if not files or args.variant != 'full' or args.batch != 32:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    val = {}
    for ph in t.get('phases', []):
        key = 'burst' if ph['phase'].startswith('burst') else 'sustained'
        val[key] = {'clock_ghz_median': ph.get('clock_ghz', {}).get('median'), 'clock_ghz_p10': ph.get('clock_ghz', {}).get('p10'),
                    'clock_ghz_p90': ph.get('clock_ghz', {}).get('p90'),
                    'cycles_per_k_step': {k: v.get('cycles_per_k_step') for k, v in ph.get('by_tile', {}).items()},
                    'images_per_s_of_the_stamped_build': ph.get('images_per_s')}
    src = {'file': os.path.relpath(files[-1], ROOT),
           'note': 'in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the K loop of every 256 x 256 / 256 x 128 tile of the '
                   'launches that do not split K, diagnostic library, same workload / pipeline / loop; a K step issues 2048 (256 x 128: 1024) '
                   'cycles of MFMAs; not collected in this run'}
    return val, src


def parity_mode_leg(args, ron_class, ron_params, weights, images, dev, detect_args, top_k, ref_dets):
    """The same workload in the arithmetic that meets north_star's float tolerance (split precision, dtype f16x3: detections
    within 1e-4 of the fp32 CPU reference), AFTER the headline's timed region and outside its clock: a second context over the
    same weights and images, `--parity-steps` timed steps through the same two-slot pipeline and the same bench_loop.
    `ref_dets`: the fp32 oracle's detections of the first images (from the cpu_baseline leg), or None."""
    import torch
    from ron_tensorflow_amd import parallel
    from ron_tensorflow_amd.metrics import detection_agreement
    from ron_tensorflow_amd.pipeline import DetectPipeline
    dtype = 'f16x3'
    if args.variant == 'ssd512':
        net = ron_class(ron_params, dtype=dtype, max_batch=args.batch, device=dev, fuse_pools=True)
    else:
        net = ron_class(ron_params, variant=args.variant, dtype=dtype, max_batch=args.batch, device=dev, fuse_pools=True,
                        head_plan=args.head_plan)
    net.load_weights(weights)
    in_flight = max(1, args.in_flight)
    pipe = DetectPipeline(net, slots=in_flight, top_k=top_k, max_queued=args.max_queued)
    steps = max(1, args.parity_steps)
    # (the GPU idled through the cpu_baseline leg: enough warm-up steps for the clocks to come back up before the timed ones)
    res = parallel.bench_loop(pipe, images, steps, PARITY_WARMUP, in_flight, detect_args, top_k, device=dev)
    dt, det = res['dt'], res['det']
    sustained = None
    if args.sustained_seconds > 0:
        sustained = sustained_leg(pipe, images, dt / steps * 1e3, args.sustained_seconds, in_flight, detect_args, top_k, args.batch,
                                  args.batch * steps / dt, device=dev)
    tflops = net.flops_per_image() * args.batch * steps / dt / 1e12
    out = {'dtype': dtype, 'images_per_s': args.batch * steps / dt, 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'warmup': PARITY_WARMUP,
           'batches_in_flight': in_flight, 'conv_stack_tflops': tflops, 'peak_tflops': PEAK_F16X3_TFLOPS,
           'roofline_frac_of_833': tflops / PEAK_F16X3_TFLOPS,
           'note': 'same weights, images, pipeline and loop as the headline, run after its timed region; three f16 MFMAs per '
                   'algorithmic product, hence the peak of 2500 / 3 TFLOP/s'}
    if sustained is not None:
        out['sustained'] = sustained
        out['sustained']['conv_stack_tflops'] = net.flops_per_image() * sustained['images_per_s'] / 1e12
        out['sustained']['roofline_frac_of_833'] = out['sustained']['conv_stack_tflops'] / PEAK_F16X3_TFLOPS
    if ref_dets is not None:
        got = det.to_lists()
        agr = [detection_agreement(got[i], ref_dets[i], tol=1e-4) for i in range(len(ref_dets))]
        nref = max(sum(a['n_ref'] for a in agr), 1)
        out['agreement'] = {'images': len(ref_dets), 'reference_detections': sum(a['n_ref'] for a in agr),
                            'reproduced': sum(a['reproduced'] * a['n_ref'] for a in agr) / nref,
                            'within_1e-4_of_reproduced': float(np.mean([a['within_tol'] for a in agr])),
                            'max_score_diff': max(a['max_score_diff'] for a in agr), 'max_box_diff': max(a['max_box_diff'] for a in agr)}
    pipe.close()
    net.close()
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from ron_tensorflow_amd import parallel
    lay = parallel.rank_layout()
    world, rank, local_rank = lay['world'], lay['rank'], lay['local_rank']
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit('for --gpus > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d '
                         '--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ...' % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # under torch.distributed.run the process group (RCCL) exists even for one rank, so that `--nproc-per-node 1` exercises
    # the same gather path as N > 1
    use_dist = lay['use_dist']
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from ron_tensorflow_amd import _lib
    from ron_tensorflow_amd.nets import nets_factory
    from ron_tensorflow_amd.weights import ssd_synthetic_weights, synthetic_images, synthetic_weights
    import ctypes as C

    # ---- the reference's call pattern (eval_ron_network.py:148-152): factory -> class -> params -> net
    ssd = args.variant == 'ssd512'
    ron_class = nets_factory.get_network('ssd_512_vgg' if ssd else 'ron_320_vgg')
    ron_params = ron_class.default_params._replace(num_classes=21)
    # the same weights on every rank: made once per node and shared through /dev/shm (12 s of RNG per rank otherwise)
    if ssd:
        weights, weights_s, weights_how = parallel.shared_host_arrays(lambda: ssd_synthetic_weights(seed=5), 'ssd512_5', local_rank, use_dist)
        net = ron_class(ron_params, dtype=args.dtype, max_batch=args.batch, device=dev, fuse_pools=True)
    else:
        # seed 1: ~4.8 k candidates, ~230 detections per image (full)
        weights, weights_s, weights_how = parallel.shared_host_arrays(lambda: synthetic_weights(args.variant, seed=1), args.variant + '_1',
                                                                      local_rank, use_dist)
        net = ron_class(ron_params, variant=args.variant, dtype=args.dtype, max_batch=args.batch, device=dev, fuse_pools=True,
                        multi_stream=args.multi_stream, head_plan=args.head_plan)
    net.load_weights(weights)
    images = torch.from_numpy(synthetic_images(args.batch, seed=lay['image_seed'], img_shape=ron_params.img_shape)).to(dev)   # resident in HBM
    top_k = 400

    # One step = one batch through ron_detect.  `--in-flight F` batches are kept in flight on F execution slots (shared
    # weights, one stream each): a step is submitted as soon as its slot's previous batch has been consumed.  The loop itself
    # (step / consume / gather / check, barriers, MAX over ranks) is parallel.bench_loop, which the world-size-2 gloo test
    # drives with a stub pipeline.
    from ron_tensorflow_amd.pipeline import DetectPipeline
    in_flight = max(1, args.in_flight)
    pipe = DetectPipeline(net, slots=in_flight, top_k=top_k, max_queued=args.max_queued)
    detect_args = dict(select_threshold=0.01, nms_threshold=0.45) if ssd else \
        dict(objectness_thres=0.03, select_threshold=0.01, nms_threshold=0.45)
    io_stream = torch.cuda.Stream(device=dev)      # the consumer (record packing + RCCL gather) has its own stream
    lib = _lib.lib()
    contexts = [slot._context() for slot in pipe.slots]

    def before_timed():
        # HIP events around every launch for PROFILED_STEPS of each slot's timed steps (each event costs ~3 us of host +
        # queue time, so not on all of them)
        for ctx in contexts:
            _lib.check(lib.ron_profile_reset(ctx))
            _lib.check(lib.ron_profile_enable(ctx, min(PROFILED_STEPS, args.steps)))

    def after_timed(_det):
        for ctx in contexts:
            _lib.check(lib.ron_profile_enable(ctx, 0))

    res = parallel.bench_loop(pipe, images, args.steps, args.warmup, in_flight, detect_args, top_k, rank=rank, world=world,
                              use_dist=use_dist, device=dev, check_gather=args.check_gather, consumer_stream=io_stream,
                              before_timed=before_timed, after_timed=after_timed)
    dt, det, gather_check = res['dt'], res['det'], res['gather_check']
    ranks_seen = None
    if use_dist:
        # who took part: the process group's size and the distinct GPUs behind its ranks (one process per GPU: they must be equal)
        mine = {'rank': rank, 'local_rank': local_rank, 'device': torch.cuda.current_device(),
                'uuid': str(getattr(torch.cuda.get_device_properties(dev), 'uuid', 'cuda:%d' % local_rank)), 'image_seed': lay['image_seed']}
        every = [None] * world
        dist.all_gather_object(every, mine)
        ranks_seen = {'world_size': dist.get_world_size(), 'distinct_devices': len({e['uuid'] for e in every}),
                      'device_of_rank': [e['device'] for e in every], 'image_seed_of_rank': [e['image_seed'] for e in every]}

    # ---- per-launch timing of the timed region (HIP events on the launch stream, inside libron_hip)
    def collect_rows(ctxs):
        rows = []
        nops = lib.ron_profile_num_ops(ctxs[0])
        for i in range(nops):
            row = None
            for ctx in ctxs:                       # the same launch on every slot: durations and counts add up
                name, is_conv, fl, ms, ln = C.c_char_p(), C.c_int(), C.c_double(), C.c_double(), C.c_int()
                ab, wb = C.c_double(), C.c_double()
                _lib.check(lib.ron_profile_get(ctx, i, C.byref(name), C.byref(is_conv), C.byref(fl), C.byref(ms), C.byref(ln),
                                               C.byref(ab), C.byref(wb)))
                if row is None:
                    row = dict(name=name.value.decode(), is_conv=bool(is_conv.value), gflop_per_image=fl.value / 1e9,
                               total_ms=0.0, launches=0, bytes_per_launch=ab.value * args.batch + wb.value)
                row['total_ms'] += ms.value
                row['launches'] += ln.value
            rows.append(row)
        return rows

    rows = collect_rows(contexts)

    # ---- the per-kernel view: PROFILED_STEPS more steps AFTER the timed region, one launch at a time (slot 0 only, nothing
    # else in flight), so that a launch's HIP-event duration is that kernel's alone
    solo_rows = rows
    if in_flight > 1:
        _lib.check(lib.ron_profile_reset(contexts[0]))
        _lib.check(lib.ron_profile_enable(contexts[0], PROFILED_STEPS))
        for _ in range(PROFILED_STEPS):
            pipe.slots[0].detect(images, top_k=top_k, **detect_args)
        torch.cuda.synchronize()
        _lib.check(lib.ron_profile_enable(contexts[0], 0))
        solo_rows = collect_rows(contexts[:1])
    # ---- the sustained leg: the same loop for >= --sustained-seconds, behind the timed region and the per-kernel steps (every rank takes
    # part; with a process group the gather stays in the loop).  It comes AFTER the solo steps: three seconds of two batches in flight leave the
    # chip in the power state of that load, and solo steps taken right behind them read 10 % slower than the same steps behind a 75 ms
    # burst (per-kernel 0.465 vs 0.514 of peak, round 6)
    sustained = None
    if args.sustained_seconds > 0:
        sustained = sustained_leg(pipe, images, dt / args.steps * 1e3, args.sustained_seconds, in_flight, detect_args, top_k,
                                  world * args.batch, world * args.batch * args.steps / dt, rank=rank, world=world,
                                  use_dist=use_dist, device=dev, consumer_stream=io_stream)
    solo_conv = [r for r in solo_rows if r['is_conv'] and r['launches'] > 0]
    solo_ms = sum(r['total_ms'] for r in solo_conv)
    solo_flop = sum(r['gflop_per_image'] * 1e9 * args.batch * r['launches'] for r in solo_conv)
    solo_launches = sum(r['launches'] for r in solo_conv)
    per_kernel_tflops = solo_flop / (solo_ms * 1e-3) / 1e12 if solo_ms > 0 else 0.0
    conv = [r for r in rows if r['is_conv'] and r['launches'] > 0]
    conv_ms = sum(r['total_ms'] for r in conv)
    conv_launches = sum(r['launches'] for r in conv)
    conv_flop = sum(r['gflop_per_image'] * 1e9 * args.batch * r['launches'] for r in conv)
    profiled_steps = max((r['launches'] for r in conv), default=0)
    conv_gflop_per_image = sum(r['gflop_per_image'] for r in rows if r['is_conv'])
    per_launch_tflops = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    if in_flight == 1:
        # one launch at a time: algorithmic FLOPs per launch / that kernel's average launch duration
        achieved = per_launch_tflops
        basis = 'conv FLOPs per launch / HIP-event duration of the launch (launches do not overlap)'
    else:
        # launches of different slots share the GPU, so a launch's duration is no longer the kernel's alone (and the
        # durations of a step add up to more than the step): charge the conv kernel with the WHOLE timed region instead
        achieved = conv_gflop_per_image * 1e9 * args.batch * args.steps / dt / 1e12
        basis = ('%d batches in flight: conv FLOPs of the timed region / timed wall time (the other kernels\' time is charged to '
                 'the conv kernel too); per-launch durations overlap, see concurrency' % in_flight)
    algo_bytes = sum(r['bytes_per_launch'] * r['launches'] for r in conv) / max(conv_launches, 1)
    traffic, traffic_source = load_traffic(args)
    mfma_busy, mfma_busy_source = load_mfma_busy(args)
    k_loop_clock, k_loop_clock_source = load_kloop_clock(args, in_flight)
    peak = {'bf16': PEAK_BF16_TFLOPS, 'fp16': PEAK_F16_TFLOPS, 'fp32': PEAK_F32_TFLOPS, 'f16x3': PEAK_F16X3_TFLOPS}[args.dtype]

    if rank == 0:
        total_images = world * args.batch * args.steps
        out = {
            'metric': 'images/sec SSD-VGG-512 inference' if ssd else 'images/sec RON-VGG16-320 inference',
            'value': total_images / dt,
            'unit': 'images/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': '%s (%s) batch=%d per GPU, synthetic %dx%dx3 inputs resident in HBM, '
                                   'forward + np_methods decode/select/top-400/NMS%s'
                                   % ({'full': 'RON-320 VGG16 ron_net', 'reducedfc': 'RON-320 reducedfc', 'ssd512': 'SSD-VGG-512'}[args.variant],
                                      args.dtype, args.batch, ron_params.img_shape[0], ron_params.img_shape[1],
                                      ', RCCL all-gather of detection records' if world > 1 else ''),
                       'batches_in_flight': in_flight,
                       'images_per_step': world * args.batch,
                       'gflop_per_image': net.flops_per_image() / 1e9,
                       'conv_stack_tflops_per_gpu': net.flops_per_image() * args.batch * args.steps / dt / 1e12,
                       'mean_detections_per_image': float(det.count.float().mean().item())},
            'roofline': {'bound': 'mfma', 'kernel': 'conv kernels: conv_igemm_kernel (+ its grouped form conv_igemm_group_kernel), conv3x3_patch_kernel and conv3x3_c64_kernel', 'achieved': achieved, 'peak': peak,
                         'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic, 'traffic_source': traffic_source,
                         'mfma_busy': mfma_busy, 'mfma_busy_source': mfma_busy_source,
                         'k_loop_clock': k_loop_clock, 'k_loop_clock_source': k_loop_clock_source,
                         'basis': basis,
                         # the per-kernel definition: conv FLOPs per launch / that launch's own duration, one launch at a
                         # time (3 steps after the timed region when several batches were in flight during it)
                         'per_kernel_tflops': per_kernel_tflops, 'per_kernel_frac': per_kernel_tflops / peak,
                         'per_kernel_avg_launch_us': solo_ms / max(solo_launches, 1) * 1e3,
                         'per_launch_tflops': per_launch_tflops,
                         'concurrency': conv_ms / max(profiled_steps, 1) / (dt / args.steps * 1e3) if in_flight > 1 else 1.0,
                         'algorithmic_bytes_per_launch': algo_bytes,
                         'avg_launch_us': conv_ms / max(conv_launches, 1) * 1e3,
                         'launches_per_step': conv_launches // max(profiled_steps, 1),
                         'profiled_steps': profiled_steps,
                         'kernel_time_share': conv_ms / max(profiled_steps, 1) / (dt / args.steps * 1e3)},
        }
        if sustained is not None:
            sustained['conv_stack_tflops_per_gpu'] = net.flops_per_image() * sustained['images_per_s'] / world / 1e12
            sustained['roofline_frac'] = sustained['conv_stack_tflops_per_gpu'] / peak
            out['sustained'] = sustained
        if gather_check is not None:
            out['gather_check'] = gather_check
        if use_dist:
            per_rank = [args.batch * args.steps / t for t in res['rank_dt']]
            out['ranks_seen'] = ranks_seen
            out['per_rank_images_per_s'] = {'min': min(per_rank), 'max': max(per_rank)}
            out['gather_ms'] = res['gather_ms']
        out['weights'] = {'synthesis_s': weights_s, 'how': weights_how + (' (shared through /dev/shm by local rank 0)' if use_dist and world > 1 else '')}
        if world == 1 and not args.no_cpu_baseline:
            from ron_tensorflow_amd.metrics import detection_agreement
            n1 = max(1, min(args.cpu_images, args.batch))
            out['cpu_baseline'], ref_dets = cpu_baseline(args.variant, weights, images.cpu().numpy(), n1, args.batch)
            # detections of the timed path (this dtype) vs the all-fp32 oracle on the same images (the agreement
            # rate that stands in for mAP while no checkpoint ships; north-star tolerance 1e-4 on the matched ones)
            got = det.to_lists()
            agr = [detection_agreement(got[i], ref_dets[i], tol=1e-4) for i in range(n1)]
            out['%s_vs_fp32_oracle_agreement' % args.dtype] = {
                'images': n1,
                'reference_detections': sum(a['n_ref'] for a in agr),
                'reproduced': sum(a['reproduced'] * a['n_ref'] for a in agr) / max(sum(a['n_ref'] for a in agr), 1),
                'within_1e-4_of_reproduced': float(np.mean([a['within_tol'] for a in agr])),
                'max_score_diff': max(a['max_score_diff'] for a in agr), 'max_box_diff': max(a['max_box_diff'] for a in agr),
                'note': 'same (class, anchor_index) pairs after NMS, first %d images of the batch; fp32 device mode reproduces >= 98 %% '
                        'with scores / boxes within 1e-4 (tests/test_gpu_forward.py)' % n1}
        if world == 1 and not use_dist and not args.no_parity_mode and args.dtype not in ('f16x3', 'fp32'):
            out['parity_mode'] = parity_mode_leg(args, ron_class, ron_params, weights, images, dev, detect_args, top_k,
                                                 ref_dets if not args.no_cpu_baseline else None)
        if args.layers:
            rows = solo_rows                         # each kernel alone
            with open(args.layers, 'w') as f:
                f.write('# per-launch timing, %s %s batch %d, %d steps (HIP events)\n' % (args.variant, args.dtype, args.batch, args.steps))
                f.write('%-28s %9s %10s %9s %9s\n' % ('launch', 'GFLOP/img', 'avg_us', 'TFLOP/s', 'share_%'))
                tot = sum(r['total_ms'] for r in rows) or 1.0
                for r in rows:
                    if r['launches'] == 0:
                        continue
                    us = r['total_ms'] / r['launches'] * 1e3
                    tf = r['gflop_per_image'] * args.batch / (us * 1e-6) / 1e3 if us > 0 else 0
                    f.write('%-28s %9.3f %10.1f %9.1f %9.2f\n' % (r['name'], r['gflop_per_image'], us, tf, 100 * r['total_ms'] / tot))
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    if gather_check == 'MISMATCH':
        raise SystemExit('gathered detection records differ from the local ones')


if __name__ == '__main__':
    main()
