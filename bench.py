#!/usr/bin/env python3
"""bench.py -- images/sec of the box-processing hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Invoked plainly with --gpus N > 1 (no WORLD_SIZE in the environment) it starts its own N ranks --
as children, before this process touches the GPU -- relays rank 0's JSON line and exits with the
children's status; with fewer than N GPUs visible it exits non-zero instead of running fewer ranks.

A "step" is ONE pass of match+encode (odbox_match_encode: IoU matching + target encoding,
dense [B,N,5] + [B,N,C+1] outputs) over one per-GPU batch of BASELINE.json configs[1]
(YOLOv3-shaped 416x416 grid, 10 647 anchors, 80 classes, 20 GT boxes, batch 64 per GPU), inputs
resident in HBM.  Consecutive steps write DIFFERENT output buffers (OUTPUT_SETS sets, 1.2 GB in
rotation), so no step's 245 MB of stores can be absorbed by the 256 MB Infinity Cache: `value` and
`roofline` are rates against HBM.  The rate with ONE output set rewritten every step (what a steady
loop that reuses its buffers sees; cache-assisted) is reported beside it as `roofline_cache_resident`.
Images are independent, so ranks hold different images and exchange nothing (weak scaling).
The same JSON line carries
  roofline            the match_encode kernel against the 8 TB/s HBM peak: algorithmic bytes / mean HIP-event
                      time per launch over the timed steps; `us_per_launch_median` = median over 10 groups of 10 steps,
  match_encode_cfg0 / _cfg2 / _cfg3   the same kernel on configs[0] (B 8), the per-GPU share of configs[2]
                      (B 32, 196 416 anchors, 2.26 GB per launch) and configs[3] (500 GT boxes; VALU-bound),
  cpu_baseline        the oracle's C restatement timed on this host's cores on a bounded sample,
  match_loss          SURVEY.md 8f row 1 on the same batch: matcher fused with the focal + Huber loss,
  decode_nms          the inference-side path (configs[4] share: 49 104 anchors, 90 classes, 64 images per GPU,
                      + one RCCL all-gather of the detections when N > 1): `value` = one call after the other, the
                      reference's call pattern; `pipelined` = two batches in flight (needs a caller written for it).
A multi-GPU run (SCALE) reads the top-level `value` (match+encode, no collective) and `decode_nms.value` (with the gather).
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from object_detection_amd import _lib, anchors as A, bndbox as BB, dist as oddist, workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (about 6.3 TB/s achievable)
OUTPUT_SETS = 5                # output buffer sets in rotation for the headline: 5 x 245 MB against a 256 MB Infinity Cache
TRAFFIC_SOURCE = ('static: profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench, '
                  'see profiles/README.md), not measured in this run')


# ------------------------------------------------------------------------------------------------
# launching the ranks
# ------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Parent side of `python bench.py --gpus N`: N children through torch.distributed.run (one rank
    per GPU), their stdout / stderr inherited (rank 0 prints the JSON line), exit status relayed.
    Nothing here initialises the GPU: torch.cuda.device_count() only counts."""
    rehearsal = os.environ.get('ODBOX_DIST_BACKEND') == 'gloo'        # several ranks on one GPU / on CPU, gloo transport
    have = torch.cuda.device_count()
    if have < n and not rehearsal:
        print('bench.py: --gpus %d but only %d GPU(s) visible; refusing to run fewer ranks' % (n, have), file=sys.stderr)
        return 3
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd).returncode


def selftest_child(args):
    """`--selftest-child`: the launch contract without any GPU work (tests/test_bench_spawn.py): join
    the group, MAX-reduce one value over the ranks, rank 0 prints the JSON line."""
    os.environ.setdefault('ODBOX_DIST_BACKEND', 'gloo')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        top = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        top = 0.0
    if args.selftest_child == 'fail' and rank == world - 1:
        return 7
    if rank == 0:
        print(json.dumps({'selftest': True, 'n_gpus': world, 'max_rank': top}))
    return 0


# ------------------------------------------------------------------------------------------------
# timing
# ------------------------------------------------------------------------------------------------
def yolo_grid_anchors(S=416):
    parts = []
    for size, stride in ((32, 8), (64, 16), (128, 32)):
        g = A.AnchorGenerator(size, [1], stride)
        parts.append(g((S // stride, S // stride, 3)))
    return torch.cat(parts, 0)


def barrier(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed(fn, steps, warmup, world):
    """W untimed + exactly K timed calls fn(i), barrier + synchronize on both sides; returns
    (max-over-ranks wall seconds, HIP-event seconds on the launch stream of this rank).  The
    library launches on torch's current stream (_lib.stream_ptr()), which is the stream these
    events are recorded on.  The warm-up goes through the same event / synchronize calls as the timed
    region (a fresh box pages code in on first use: 25 ms once landed inside a 10 ms timed region),
    and the garbage collector stays out of it."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    gc.collect()              # before the warm-up, not between it and the timed region: a collection idles the GPU for
    gc.disable()              # tens of ms, and the first of only K = 20 timed steps then run while its clocks ramp up
    try:
        e0.record()
        for i in range(warmup):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        e0.elapsed_time(e1)
        barrier(world)
        t0 = time.perf_counter()
        e0.record()
        for i in range(steps):
            fn(warmup + i)
        e1.record()
        torch.cuda.synchronize()
        barrier(world)
        wall = time.perf_counter() - t0
    finally:
        gc.enable()
    return max_over_ranks(wall, world), e0.elapsed_time(e1) * 1e-3


def per_step_median(fn, groups=10, per_group=10):
    """Median (and (min, max)) over `groups` of the mean HIP-event time of `per_group` consecutive steps: a
    single hiccup cannot move the figure, and the event marker between two groups
    (about 2.5 us on the stream) is spread over per_group steps.  Seconds per step."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
    torch.cuda.synchronize()
    gc.disable()
    try:
        for g in range(groups):
            ev[g].record()
            for i in range(per_group):
                fn(g * per_group + i)
        ev[groups].record()
        torch.cuda.synchronize()
    finally:
        gc.enable()
    ts = [ev[g].elapsed_time(ev[g + 1]) * 1e-3 / per_group for g in range(groups)]
    return statistics.median(ts), (min(ts), max(ts))


PLAN_KERNELS = {0: 'match_encode_kernel<false, false>', 1: 'match_encode_kernel<true, false>',
                2: 'match_encode_sweep_kernel'}


def match_encode_kernel_name(lib, B, N, G, C):
    """The kernel odbox_match_encode launches for this shape under the output policy in force (odbox_match_encode_plan)."""
    return PLAN_KERNELS.get(lib.odbox_match_encode_plan(B, N, G, C, 1, 0), 'match_encode_kernel')


def hbm_roofline(kernel, byts, mean_s, median_s, traffic_key=None, spread=None, **extra):
    """`achieved` / `frac` come from the MEDIAN over groups of steps (per_step_median: 10 groups of 10 by default) --
    the driver's --steps 20 are 0.9 ms of GPU time for the headline, too short a region for a stable mean -- and the
    mean over exactly the K timed steps sits beside it (`us_per_launch_timed_steps`, `frac_timed_steps`) with the
    spread of the group means."""
    ach = byts / median_s / 1e9
    r = {'bound': 'hbm', 'kernel': kernel, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
         'frac': round(ach / HBM_PEAK_GBS, 4), 'bytes_per_launch': byts, 'us_per_launch': round(median_s * 1e6, 3),
         'us_per_launch_median': round(median_s * 1e6, 3),
         'us_per_launch_timed_steps': round(mean_s * 1e6, 3),
         'frac_timed_steps': round(byts / mean_s / 1e9 / HBM_PEAK_GBS, 4),
         'traffic': load_traffic(traffic_key) if traffic_key else None}
    if spread:
        r['us_per_launch_group_min_max'] = [round(spread[0] * 1e6, 3), round(spread[1] * 1e6, 3)]
    if traffic_key:
        r['traffic_source'] = TRAFFIC_SOURCE
    r.update(extra)
    return r


def cpu_baseline_match(anchors_np, gt, lab, C, S, budget_s=6.0):
    """The oracle's C restatement (kind "port") on this host: same cfg-2 batch, repeated until
    about `budget_s` seconds per thread setting."""
    from oracle import c_oracle
    out = {}
    B = gt.shape[0]
    for name, threads in (('multi', min(16, os.cpu_count() or 1)), ('single', 1)):
        c_oracle.match_encode(anchors_np, (S, S), gt[:8], lab[:8], C, threads=threads)     # warm
        n, t0 = 0, time.perf_counter()
        while True:
            c_oracle.match_encode(anchors_np, (S, S), gt, lab, C, threads=threads)
            n += B
            el = time.perf_counter() - t0
            if el > budget_s:
                break
        out[name] = (n / el, threads, n)
    return out


def cpu_baseline_nms(anchors_np, reg_np, sc_np, S, budget_s=4.0):
    """The oracle's C decode + NMS on a bounded sample of the same cfg-5 batch (16 images per pass,
    repeated for about `budget_s` seconds per thread setting)."""
    from oracle import c_oracle
    out = {}
    n_img = min(16, reg_np.shape[0])
    reg_np, sc_np = reg_np[:n_img], sc_np[:n_img]
    for name, threads in (('multi', min(16, os.cpu_count() or 1)), ('single', 1)):
        c_oracle.decode_nms(anchors_np, reg_np[:2], sc_np[:2], (S, S), threads=threads)       # warm
        n, t0 = 0, time.perf_counter()
        while True:
            c_oracle.decode_nms(anchors_np, reg_np, sc_np, (S, S), threads=threads)
            n += n_img
            el = time.perf_counter() - t0
            if el > budget_s:
                break
        out[name] = (n / el, threads, n)
    return out


def match_encode_case(lib, dev, stream, cfg, B, anchors, seed, sets, pad_to=None):
    """Device buffers + a step(i) closure of odbox_match_encode on BASELINE config `cfg` with batch B,
    writing output set i % sets.  pad_to: GT rows padded with -1 up to that many (detection.py:246-266 pads to 64)."""
    c = workloads.CONFIGS[cfg]
    S, C = c['S'], c['C']
    N = anchors.shape[0]
    gt_np, lab_np = workloads.match_inputs(cfg, batch=B, seed=seed, pad_to=pad_to)
    gt = torch.as_tensor(gt_np).to(dev)
    lab = torch.as_tensor(lab_np).to(dev)
    G = gt.shape[1]
    outs = [(torch.empty((B, N, 5), dtype=torch.float32, device=dev),
             torch.empty((B, N, C + 1), dtype=torch.float32, device=dev)) for _ in range(sets)]

    def step(i):
        reg, labels = outs[i % sets]
        rc = lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C,
                                    float(S), float(S), 0.4, 0.5, reg.data_ptr(), labels.data_ptr(),
                                    None, None, None, None, stream)
        if rc:
            _lib.check(rc, 'odbox_match_encode')
    return step, dict(S=S, B=B, N=N, G=G, C=C, gt_np=gt_np, lab_np=lab_np, gt=gt, lab=lab, outs=outs,
                      bytes=B * workloads.match_bytes_per_image(N, G, C))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--nms-steps', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-nms', action='store_true')
    ap.add_argument('--no-loss', action='store_true')
    ap.add_argument('--no-nms-pipeline', action='store_true',
                    help='decode+NMS one batch at a time only (clean per-kernel durations under rocprofv3: with two '
                         'batches in flight the tail kernels of one batch are stretched over the scan of the other)')
    ap.add_argument('--no-configs', action='store_true', help='skip the configs[0] / [2] / [3] sub-objects')
    ap.add_argument('--loss-steps', type=int, default=50)
    ap.add_argument('--selftest-child', nargs='?', const='ok', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    # ---- the launch contract, decided before anything touches the GPU ---------------------------
    env_world = os.environ.get('WORLD_SIZE')
    if args.gpus < 1:
        print('bench.py: --gpus must be >= 1', file=sys.stderr)
        return 2
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])
    if env_world is not None and int(env_world) != args.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (or let bench.py start them: '
              'run it without torch.distributed.run)' % (args.gpus, env_world), file=sys.stderr)
        return 2
    if args.selftest_child:
        return selftest_child(args)

    rank, world = oddist.init_from_env()
    lib = _lib.load()
    dev = _lib.require_gpu()
    stream = _lib.stream_ptr()

    # ---------------- match + encode, configs[1] ------------------------------------------
    anchors = yolo_grid_anchors(workloads.CONFIGS[2]['S'])
    assert anchors.shape[0] == workloads.num_anchors(2)
    step, w = match_encode_case(lib, dev, stream, 2, workloads.CONFIGS[2]['B'], anchors, 1002 + 7919 * rank, OUTPUT_SETS)
    S, B, N, G, C = w['S'], w['B'], w['N'], w['G'], w['C']
    # outputs that nothing reads back soon are written around the caches: the caller says so
    # (odbox_set_output_policy, anchors.set_output_policy('stream')); the library's size rule alone
    # (`auto`) would send 221 MB of labels through them
    lib.odbox_set_output_policy(2)
    k_stream = match_encode_kernel_name(lib, B, N, G, C)
    wall, ev = timed(step, args.steps, args.warmup, world)
    med, spr = per_step_median(step)
    img_s = world * B * args.steps / wall
    lib.odbox_set_output_policy(0)
    _, ev0 = timed(step, args.steps, args.warmup, 1)                   # the same rotation under the default policy
    med0, spr0 = per_step_median(step)
    k_auto = PLAN_KERNELS.get(lib.odbox_match_encode_last_plan(), 'match_encode_kernel')      # what AUTO launched (a function of the shape)
    # the same launch rewriting ONE output set every step, through the caches (policy 'cache': the caller knows its
    # buffer is still in the Infinity Cache): its 245 MB fit there
    lib.odbox_set_output_policy(1)
    _, ev1 = timed(lambda i: step(0), args.steps, args.warmup, 1)
    med1, spr1 = per_step_median(lambda i: step(0))
    k_resident = PLAN_KERNELS.get(lib.odbox_match_encode_last_plan(), 'match_encode_kernel')
    lib.odbox_set_output_policy(0)

    result = {
        'metric': 'images/sec match+encode (IoU matching + target encoding)',
        'value': round(img_s, 1),
        'unit': 'images/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(wall / args.steps * 1e3, 5),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': 'configs[1]: YOLOv3-shaped 416x416 grid, 3 strides x 3 anchors = %d anchors, '
                               '%d classes, %d GT boxes/img, batch %d per GPU, dense [B,N,5]+[B,N,C+1] outputs, '
                               '%d output sets in rotation (%.2f GB) so that no step is absorbed by the Infinity Cache, '
                               "output policy 'stream' (targets nothing reads back soon)"
                               % (N, C, G, B, OUTPUT_SETS, OUTPUT_SETS * B * N * (C + 6) * 4 / 1e9),
                   'global_batch': world * B, 'anchors': N, 'classes': C, 'gt_per_image': G,
                   'parallelism': 'images sharded over %d GPU(s), no collective' % world},
        'roofline': hbm_roofline(k_stream, w['bytes'], ev / args.steps, med, 'match_encode_kernel_rotating', spr,
                                 output_sets=OUTPUT_SETS, output_policy='stream'),
        'roofline_auto_policy': hbm_roofline(
            k_auto, w['bytes'], ev0 / args.steps, med0, None, spr0, output_sets=OUTPUT_SETS, output_policy='auto',
            note='the same rotation under the default policy: what a caller that says nothing about its consumer gets '
                 '(AUTO is a function of the shape and assumes a cold buffer: the sweep for this shape, on every run)'),
        'roofline_cache_resident': hbm_roofline(
            k_resident, w['bytes'], ev1 / args.steps, med1, 'match_encode_kernel', spr1, output_policy='cache',
            note="one output set rewritten every step under policy 'cache': the 245 MB of stores fit the 256 MB "
                 'Infinity Cache, so this is a cache-assisted store rate, not an HBM fraction'),
    }
    gt_np, lab_np = w['gt_np'], w['lab_np']
    gt, lab = w['gt'], w['lab']
    del w, step

    # ---------------- the same kernel on configs[0], configs[2] share, configs[3] ------------
    if not args.no_configs:
        ksteps = max(10, min(args.steps, 30))
        retina512 = A.generate_anchors(A.AnchorsConfig(), 512)
        for key, cfg, Bc, anch, sets, label in (
                ('match_encode_cfg1_padded64', 2, 64, anchors, OUTPUT_SETS,
                 'the headline batch as the reference feeds it: GT rows padded with -1 to MAX_ITEMS_IN_IMAGE = 64 '
                 '(detection.py:246-266), 20 of them boxes; %d output sets in rotation' % OUTPUT_SETS),
                ('match_encode_cfg0', 1, 8, retina512, 8,
                 'configs[0]: EfficientDet-D0 512x512, 49 104 anchors, 80 classes, 20 GT, batch 8; 8 output sets in rotation'),
                ('match_encode_cfg2', 3, 32, None, 1,
                 'configs[2] per-GPU share: 1024x1024, 196 416 anchors, 80 classes, 20 GT, batch 32 of 256 (2.26 GB per launch)'),
                ('match_encode_cfg3', 4, 128, retina512, 2,
                 'configs[3]: dense crowd, 500 GT boxes/img vs 49 104 anchors, 1 class, batch 128')):
            if anch is None:
                anch = A.generate_anchors(A.AnchorsConfig(), workloads.CONFIGS[cfg]['S'])
            stepc, wc = match_encode_case(lib, dev, stream, cfg, Bc, anch, 1000 + cfg + 7919 * rank, sets,
                                          pad_to=64 if key.endswith('padded64') else None)
            lib.odbox_set_output_policy(2 if sets > 2 else 0)           # rotating (cold) outputs: written around the caches
            k_cfg = match_encode_kernel_name(lib, Bc, wc['N'], wc['G'], wc['C'])
            wl, el = timed(stepc, ksteps, 3, world)
            medc, sprc = per_step_median(stepc, per_group=5)
            lib.odbox_set_output_policy(0)
            sub = {'value': round(world * Bc * ksteps / wl, 1), 'unit': 'images/s', 'steps': ksteps,
                   'config': {'workload': label}}
            if cfg == 4:
                # VALU-bound: 500 GT rows per anchor.  The kernel proves most pairs zero without
                # evaluating them, so "pairs / peak rate" is not a fraction of anything; the defensible figure is
                # how busy the VALUs are over the launch (SQ counters, collected in their own PMC passes).
                # `frac`: the matcher's EVALUATED work against the FMA-free fp32 VALU rate, from this run's time.
                # Pairs per launch are a constant of the seeded workload (tools/count_pairs.py, a counting build);
                # 21.25 issue slots per pair is the four-row loop's disassembly (69 full-rate VALU instructions + 4
                # quarter-rate v_rcp_f32 per four rows).  39.3 T lane-ops/s = 256 CUs x 64 lanes x 2.4 GHz (no FMA: the
                # path may not contract); the packed ceiling is twice that.  SURVEY.md 8d's nominal figure (every one of
                # the 3.1 G pairs at 16 ops) exceeds 1 because nine pairs in ten are proven zero, not evaluated.
                pairs = load_static('match_encode_cfg3_pairs') or {}
                t_med = medc
                peak = 256 * 64 * 2.4e9
                ev_slots = pairs.get('pairs_evaluated_per_launch', 0) * pairs.get('valu_slots_per_pair', 0.0)
                cull_slots = pairs.get('cull_steps_per_launch', 0) * 64 * pairs.get('valu_slots_per_cull_step', 0)
                nominal = Bc * wc['N'] * wc['G'] * 16
                sub['roofline'] = {'bound': 'valu', 'kernel': k_cfg, 'us_per_launch': round(medc * 1e6, 2),
                                   'us_per_launch_median': round(medc * 1e6, 2),
                                   'us_per_launch_timed_steps': round(el / ksteps * 1e6, 2),
                                   'us_per_launch_group_min_max': [round(sprc[0] * 1e6, 2), round(sprc[1] * 1e6, 2)],
                                   'bytes_per_launch': wc['bytes'],
                                   'hbm_frac': round(wc['bytes'] / t_med / 1e9 / HBM_PEAK_GBS, 4),
                                   'unit': 'T lane-ops/s', 'peak': round(peak / 1e12, 1),
                                   'achieved': round(ev_slots / t_med / 1e12, 2) if ev_slots else None,
                                   'frac': round(ev_slots / t_med / peak, 4) if ev_slots else None,
                                   'frac_with_work_list_build': round((ev_slots + cull_slots) / t_med / peak, 4) if ev_slots else None,
                                   'pairs_evaluated_per_launch': pairs.get('pairs_evaluated_per_launch'),
                                   'pairs_evaluated_share': round(pairs.get('pairs_evaluated_per_launch', 0) / (Bc * wc['N'] * wc['G']), 4),
                                   'pairs_source': 'static: ' + str(pairs.get('source')),
                                   'nominal_frac_39T': round(nominal / t_med / peak, 3),
                                   'nominal_frac_79T_packed': round(nominal / t_med / (2 * peak), 3),
                                   'nominal_note': 'SURVEY.md 8d: B x N x G x 16 ops over the launch time; above 1 because the '
                                                   'matcher proves most pairs zero without evaluating them (exact, tested)',
                                   'valu_busy_static': load_static('match_encode_cfg3_valu_busy'),
                                   'valu_busy_source': 'static: VALU-busy cycles / launch cycles from the SQ counter passes '
                                                       '(tools/pmc_cfg.sh 4, profiles/), not measured in this run and not `frac`'}
            else:
                sub['roofline'] = hbm_roofline(k_cfg, wc['bytes'], el / ksteps, medc, None, sprc, output_sets=sets,
                                               output_policy='stream' if sets > 2 else 'auto')
            result[key] = sub
            del stepc, wc
        torch.cuda.empty_cache()

    # ---------------- fused matcher + loss on the same batch ----------
    if not args.no_loss:
        gen = torch.Generator(device=dev).manual_seed(4242 + rank)
        pb = torch.randn((B, N, 4), device=dev, generator=gen)
        ps = torch.rand((B, N, C), device=dev, generator=gen)
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        gb, gs = torch.empty_like(pb), torch.empty_like(ps)
        ws = torch.empty(lib.odbox_match_loss_workspace(B, N), dtype=torch.uint8, device=dev)
        result['match_loss'] = {}
        common = (anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(S), float(S), 0.4, 0.5)
        scale2 = torch.tensor([0.5, 0.25], device=dev)
        hb, hs = pb.half(), ps.half()
        ghb, ghs = torch.empty_like(hb), torch.empty_like(hs)

        def fwd_step(i):
            _lib.check(lib.odbox_match_loss(*common, pb.data_ptr(), ps.data_ptr(), 1.5, 0.25, 1.0, out4.data_ptr(), None, None,
                                            ws.data_ptr(), ws.numel(), stream), 'forward')

        def fwd_bwd_step(i):
            _lib.check(lib.odbox_match_loss(*common, pb.data_ptr(), ps.data_ptr(), 1.5, 0.25, 1.0, out4.data_ptr(), gb.data_ptr(),
                                            gs.data_ptr(), ws.data_ptr(), ws.numel(), stream), 'forward_backward')

        def bwd_step(i):
            _lib.check(lib.odbox_match_loss_backward(*common, pb.data_ptr(), ps.data_ptr(), 1.5, 0.25, 1.0,
                                                     scale2[0:1].data_ptr(), scale2[1:2].data_ptr(), out4[2:3].data_ptr(), gb.data_ptr(), gs.data_ptr(), stream), 'backward')

        def fwd16_step(i):
            _lib.check(lib.odbox_match_loss_f16(*common, hb.data_ptr(), hs.data_ptr(), 1.5, 0.25, 1.0, out4.data_ptr(),
                                                None, None, ws.data_ptr(), ws.numel(), stream), 'forward_f16')

        def bwd16_step(i):
            _lib.check(lib.odbox_match_loss_backward_f16(*common, hb.data_ptr(), hs.data_ptr(), 1.5, 0.25, 1.0,
                                                         scale2[0:1].data_ptr(), scale2[1:2].data_ptr(), out4[2:3].data_ptr(), ghb.data_ptr(), ghs.data_ptr(), stream),
                       'backward_f16')
        fixed = B * (N * 16 + 20 * G)                                          # anchors + ground truth, always fp32
        full = B * (N * (32 + 4 * C) + 20 * G)                                 # anchors + predictions
        for name, fn, byt, tkey in (('forward', fwd_step, full, 'match_loss_kernel_fwd'),
                                    ('forward_backward', fwd_bwd_step, 2 * full, 'match_loss_kernel_fwd_bwd'),
                                    ('backward', bwd_step, fixed + 2 * B * N * (16 + 4 * C), None),
                                    ('forward_f16', fwd16_step, fixed + B * N * (8 + 2 * C), 'match_loss_kernel_f16_fwd'),
                                    ('backward_f16', bwd16_step, fixed + 2 * B * N * (8 + 2 * C), 'match_loss_kernel_f16_bwd')):
            wl, el = timed(fn, args.loss_steps, 50, world)
            medl, sprl = per_step_median(fn)
            result['match_loss'][name] = {
                'value': round(world * B * args.loss_steps / wl, 1), 'unit': 'images/s',
                'us_per_step': round(el / args.loss_steps * 1e6, 2), 'steps': args.loss_steps,
                'roofline': hbm_roofline('match_loss_kernel (+ its reduction)', byt, el / args.loss_steps, medl, tkey, sprl)}
        del pb, ps, gb, gs, hb, hs, ghb, ghs

    # ---------------- decode + NMS, configs[4] per-GPU share -------------------------------
    nms_cpu_sample = None
    if not args.no_nms:
        S5, B5, C5 = 512, 64, 90
        anchors5 = A.generate_anchors(A.AnchorsConfig(), S5)
        N5 = anchors5.shape[0]
        reg5_np, sc5_np = workloads.nms_inputs(N5, C=C5, batch=B5, seed=1005 + 7919 * rank,
                                               p3p4_anchors=9 * (64 * 64 + 32 * 32), distinct=B5)
        reg5 = torch.as_tensor(reg5_np).to(dev)
        sc5 = torch.as_tensor(sc5_np).to(dev)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            nms_cpu_sample = (anchors5.cpu().numpy(), reg5_np[:16].copy(), sc5_np[:16].copy(), S5)   # timed last
        del reg5_np, sc5_np
        out = {}

        def nms_step(i):
            # what the reference's loop does (evaluate.py:83-85, encoder.py:200-214): one batch after the other
            b, s, l = BB.decode_and_nms(anchors5, reg5, sc5, (S5, S5), pre_nms_topk=5000)
            out['det'] = oddist.gather_detections(b, s, l, batch=None)

        wall5, ev5 = timed(nms_step, args.nms_steps, 40, world)      # (8 ms of warm-up: the clocks need about 30 calls after the host-side set-up)
        med5, spr5 = per_step_median(nms_step, per_group=4)
        bytes5 = B5 * workloads.nms_bytes_per_image(N5, C5)
        digest = hashlib.sha1()
        for t in out['det']:
            digest.update(t.cpu().numpy().tobytes())
        result['decode_nms'] = {
            'metric': 'images/sec decode+clip+per-class NMS+top-k, one call after the other'
                      + (' + RCCL all-gather of the detections' if world > 1 else ''),
            'value': round(world * B5 * args.nms_steps / wall5, 1), 'unit': 'images/s',
            'ms_per_step': round(wall5 / args.nms_steps * 1e3, 4), 'steps': args.nms_steps,
            'config': {'workload': 'configs[4] per-GPU share: %d anchors, %d classes, batch %d per GPU (%d distinct synthetic '
                                   'images), score 0.3, IoU 0.45, max_ret 100, pre-NMS top-k 5000 (~4500 candidates/img: it '
                                   'never binds); the 1.13 GB score tensor is 4x the Infinity Cache; ONE batch at a time '
                                   'through bndbox.decode_and_nms -- the call pattern of evaluate.py:83-85 -- so `value` is the '
                                   'drop-in rate; `pipelined` is what a caller that keeps two batches in flight gets'
                                   % (N5, C5, B5, B5),
                       'global_batch': world * B5},
            'roofline': hbm_roofline('whole decode+NMS call: score_scan (HBM-bound), then tier_tail_kernel (per-image tier, '
                                     'partition, NMS and top-k in one block; latency-bound) and the pass-2 launch', bytes5, ev5 / args.nms_steps, med5, 'score_scan_kernel', spr5,
                                     frac_scores_only=round(B5 * N5 * C5 * 4 / med5 / 1e9 / HBM_PEAK_GBS, 4)),
            'detections_per_image_mean': float((out['det'][1] > 0).sum().item()) / (world * B5),
            'detections_sha1': digest.hexdigest(),
        }

    if not args.no_nms and not args.no_nms_pipeline:
        # two batches in flight (bndbox.DecodeNmsPipeline): the scan of batch k+1 runs beside the tail of batch k
        pipe = BB.DecodeNmsPipeline(depth=2)
        ksteps = max(args.nms_steps, 2) * 3

        def pipelined(k):
            prev = None
            for _ in range(k):
                cur = pipe.submit(anchors5, reg5, sc5, (S5, S5), pre_nms_topk=5000, inputs_ready=True)
                if prev is not None:
                    out['det2'] = oddist.gather_detections(*prev.result(), batch=None)
                prev = cur
            out['det2'] = oddist.gather_detections(*prev.result(), batch=None)

        def timed_pipeline(fn):
            """Wall clock of fn(ksteps), max over ranks, median of 3 runs (the work is on several streams: no single
            event pair brackets it)."""
            fn(6)
            torch.cuda.synchronize()
            walls = []
            gc.collect()
            gc.disable()
            try:
                for _ in range(3):
                    barrier(world)
                    t0 = time.perf_counter()
                    fn(ksteps)
                    torch.cuda.synchronize()
                    barrier(world)
                    walls.append(max_over_ranks(time.perf_counter() - t0, world))
            finally:
                gc.enable()
            return statistics.median(walls)

        wall2 = timed_pipeline(pipelined)
        same = all(torch.equal(a, b) for a, b in zip(out['det'], out['det2']))
        result['decode_nms']['pipelined'] = {
            'note': 'TWO batches in flight on two streams (bndbox.DecodeNmsPipeline; no counterpart in the reference, whose '
                    'loop decodes one batch at a time): needs a caller written around submit() / result(), so it is not '
                    '`value`',
            'value': round(world * B5 * ksteps / wall2, 1), 'unit': 'images/s',
            'ms_per_step': round(wall2 / ksteps * 1e3, 4), 'steps': ksteps,
            'frac': round(bytes5 / (wall2 / ksteps) / 1e9 / HBM_PEAK_GBS, 4),
            'timing': 'wall clock, median of 3 runs of %d steps (work is on two streams: no single event pair brackets it)' % ksteps,
            'equals_single_calls': bool(same)}
        # the same batch handed over as bfloat16 model outputs (odbox_decode_nms_mixed): half the score bytes, the
        # detections of the widened tensors; reported beside the fp32 figures, never in their place
        sc16, reg16 = sc5.to(torch.bfloat16), reg5.to(torch.bfloat16)
        bytes16 = B5 * (N5 * (16 + 8 + 2 * C5) + 24 * 100)

        def nms16_step(i):
            b, s, l = BB.decode_and_nms(anchors5, reg16, sc16, (S5, S5), pre_nms_topk=5000)
            out['det16'] = oddist.gather_detections(b, s, l, batch=None)

        pipe3 = BB.DecodeNmsPipeline(depth=3)       # the 16-bit scan is shorter than the tail beside it: three batches in flight

        def pipelined16(k):
            pending = []
            for _ in range(k):
                pending.append(pipe3.submit(anchors5, reg16, sc16, (S5, S5), pre_nms_topk=5000, inputs_ready=True))
                if len(pending) == 3:
                    out['det16'] = oddist.gather_detections(*pending.pop(0).result(), batch=None)
            for p in pending:
                out['det16'] = oddist.gather_detections(*p.result(), batch=None)

        wall16, ev16 = timed(nms16_step, args.nms_steps, 20, world)
        med16, spr16 = per_step_median(nms16_step, per_group=4)
        wall16p = timed_pipeline(pipelined16)
        want16 = BB.decode_and_nms(anchors5, reg16.float(), sc16.float(), (S5, S5), pre_nms_topk=5000)
        got16 = BB.decode_and_nms(anchors5, reg16, sc16, (S5, S5), pre_nms_topk=5000)
        result['decode_nms']['bf16_model_outputs'] = {
            'note': 'regressors and class scores as bfloat16 (what a PyTorch-ROCm model hands over under autocast), '
                    'widened exactly where they are read; not the reference precision, so not part of `value`',
            'single_stream': {'value': round(world * B5 * args.nms_steps / wall16, 1), 'unit': 'images/s',
                              'ms_per_step': round(wall16 / args.nms_steps * 1e3, 4),
                              'roofline': hbm_roofline('whole decode+NMS call, 16-bit score scan', bytes16,
                                                       ev16 / args.nms_steps, med16, None, spr16)},
            'pipelined': {'value': round(world * B5 * ksteps / wall16p, 1), 'unit': 'images/s', 'batches_in_flight': 3,
                          'ms_per_step': round(wall16p / ksteps * 1e3, 4),
                          'frac': round(bytes16 / (wall16p / ksteps) / 1e9 / HBM_PEAK_GBS, 4)},
            'equals_fp32_path_on_widened_inputs': bool(all(torch.equal(a, b) for a, b in zip(got16, want16))),
        }
        del sc16, reg16

    # ---------------- CPU baselines (rank 0, N = 1 only), LAST: everything above is GPU work back to back --------
    if nms_cpu_sample is not None:
        cb5 = cpu_baseline_nms(*nms_cpu_sample)
        result['decode_nms']['cpu_baseline'] = {
            'value': round(cb5['multi'][0], 1), 'unit': 'images/s', 'cores': cb5['multi'][1], 'kind': 'port',
            'sample': 'oracle/odbox_oracle.c decode + per-class NMS + top-k on the first 16 images of the same '
                      'batch, repeated to %d images, OpenMP over images' % cb5['multi'][2],
            'value_1core': round(cb5['single'][0], 1)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline_match(anchors.cpu().numpy(), gt_np, lab_np, C, S)
        result['cpu_baseline'] = {
            'value': round(cb['multi'][0], 1), 'unit': 'images/s', 'cores': cb['multi'][1], 'kind': 'port',
            'sample': 'oracle/odbox_oracle.c (CPU restatement with TF semantics, not TensorFlow) on the same '
                      'cfg-2 batch of %d images, repeated to %d images, OpenMP over images' % (B, cb['multi'][2]),
            'value_1core': round(cb['single'][0], 1),
            'host_cpus': os.cpu_count(),
        }
    elif rank == 0:
        result['cpu_baseline'] = None

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def load_static(key, field=None):
    """A figure from the PMC passes (profiles/pmc_traffic.json, written from the rocprofv3 --pmc runs of
    profiles/README.md); null until collected."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            v = json.load(f).get(key)
        return v.get(field) if field and isinstance(v, dict) else v
    except (OSError, ValueError):
        return None


def load_traffic(kernel):
    """HBM bytes per launch (TCC FETCH_SIZE x 2 for gfx950's wide reads + WRITE_SIZE)."""
    return load_static(kernel, 'hbm_bytes_per_launch')


if __name__ == '__main__':
    sys.exit(main())
