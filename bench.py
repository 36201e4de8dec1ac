#!/usr/bin/env python3
"""bench.py -- images/sec of the box-processing hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE pass of match+encode (odbox_match_encode: IoU matching + target encoding,
dense [B,N,5] + [B,N,C+1] outputs) over one per-GPU batch of BASELINE.json configs[1]
(YOLOv3-shaped 416x416 grid, 10 647 anchors, 80 classes, 20 GT boxes, batch 64 per GPU), inputs
resident in HBM.  Images are independent, so ranks hold different images and exchange nothing
(weak scaling).  The same JSON line carries
  roofline      the match_encode kernel against the 8 TB/s HBM peak (algorithmic bytes / HIP-event time),
  cpu_baseline  the oracle's C restatement timed on this host's cores on a bounded sample,
  match_loss    SURVEY.md 8f row 1 on the same batch: matcher fused with the focal + Huber loss
                (odbox_match_loss; the dense targets are never written), forward and forward+gradients,
  decode_nms    the inference-side path (configs[4]: 49 104 anchors, 90 classes, 64 images per GPU,
                + one RCCL all-gather of the detections when N > 1), reported beside it.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from object_detection_amd import _lib, anchors as A, bndbox as BB, dist as oddist, workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (about 6.3 TB/s achievable)


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
    """W untimed + exactly K timed calls, barrier + synchronize on both sides; returns
    (max-over-ranks wall seconds, HIP-event seconds on the launch stream of this rank)."""
    for _ in range(warmup):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    barrier(world)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    barrier(world)
    wall = time.perf_counter() - t0
    return max_over_ranks(wall, world), e0.elapsed_time(e1) * 1e-3


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--nms-steps', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-nms', action='store_true')
    ap.add_argument('--no-loss', action='store_true')
    ap.add_argument('--loss-steps', type=int, default=50)
    args = ap.parse_args()

    rank, world = oddist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print('warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    lib = _lib.load()
    dev = _lib.require_gpu()
    stream = _lib.stream_ptr()

    # ---------------- match + encode, configs[1] ------------------------------------------
    cfg = workloads.CONFIGS[2]
    S, B, G, C = cfg['S'], cfg['B'], cfg['G'], cfg['C']
    anchors = yolo_grid_anchors(S)
    N = anchors.shape[0]
    assert N == workloads.num_anchors(2)
    gt_np, lab_np = workloads.match_inputs(2, seed=1002 + 7919 * rank)
    gt = torch.as_tensor(gt_np).to(dev)
    lab = torch.as_tensor(lab_np).to(dev)
    reg = torch.empty((B, N, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((B, N, C + 1), dtype=torch.float32, device=dev)

    def step():
        rc = lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C,
                                    float(S), float(S), 0.4, 0.5, reg.data_ptr(), labels.data_ptr(),
                                    None, None, None, stream)
        if rc:
            _lib.check(rc, 'odbox_match_encode')

    wall, ev = timed(step, args.steps, args.warmup, world)
    img_s = world * B * args.steps / wall
    bytes_per_launch = B * workloads.match_bytes_per_image(N, G, C)
    kernel_s = ev / args.steps
    achieved = bytes_per_launch / kernel_s / 1e9

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
                               '%d classes, %d GT boxes/img, batch %d per GPU, dense [B,N,5]+[B,N,C+1] outputs'
                               % (N, C, G, B),
                   'global_batch': world * B, 'anchors': N, 'classes': C, 'gt_per_image': G,
                   'parallelism': 'images sharded over %d GPU(s), no collective' % world},
        'roofline': {'bound': 'hbm', 'kernel': 'match_encode_kernel',
                     'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 4),
                     'bytes_per_launch': bytes_per_launch, 'us_per_launch': round(kernel_s * 1e6, 3),
                     'traffic': load_traffic('match_encode_kernel')},
    }

    # ---------------- fused matcher + loss on the same batch ----------
    if not args.no_loss:
        gen = torch.Generator(device=dev).manual_seed(4242 + rank)
        pb = torch.randn((B, N, 4), device=dev, generator=gen)
        ps = torch.rand((B, N, C), device=dev, generator=gen)
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        gb, gs = torch.empty_like(pb), torch.empty_like(ps)
        ws = torch.empty(lib.odbox_match_loss_workspace(B, N), dtype=torch.uint8, device=dev)
        result['match_loss'] = {}
        for name, g1, g2 in (('forward', None, None), ('forward_backward', gb, gs)):
            def loss_step():
                rc = lib.odbox_match_loss(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(S),
                                          float(S), 0.4, 0.5, pb.data_ptr(), ps.data_ptr(), 1.5, 0.25, 1.0,
                                          out4.data_ptr(), None if g1 is None else g1.data_ptr(),
                                          None if g2 is None else g2.data_ptr(), ws.data_ptr(), ws.numel(), stream)
                if rc:
                    _lib.check(rc, 'odbox_match_loss')
            wl, el = timed(loss_step, args.loss_steps, 5, world)
            byt = B * (N * (32 + 4 * C) + 20 * G) * (1 if g1 is None else 2)   # anchors + predictions (+ gradients)
            ach = byt / (el / args.loss_steps) / 1e9
            result['match_loss'][name] = {
                'value': round(world * B * args.loss_steps / wl, 1), 'unit': 'images/s',
                'us_per_step': round(el / args.loss_steps * 1e6, 2), 'steps': args.loss_steps,
                'roofline': {'bound': 'hbm', 'kernel': 'match_loss_kernel + loss_reduce_kernel', 'achieved': round(ach, 1),
                             'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                             'bytes_per_launch': byt,
                             'traffic': load_traffic('match_loss_kernel_fwd' if g1 is None else 'match_loss_kernel_fwd_bwd')}}
        # the autograd backward launch (gradients only, scaled on the device) and the fp16 variants
        scale2 = torch.tensor([0.5, 0.25], device=dev)
        hb, hs = pb.half(), ps.half()
        ghb, ghs = torch.empty_like(hb), torch.empty_like(hs)
        common = (anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(S), float(S), 0.4, 0.5)

        def bwd_step():
            _lib.check(lib.odbox_match_loss_backward(*common, pb.data_ptr(), ps.data_ptr(), 1.5, 0.25, 1.0,
                                                     scale2[0:1].data_ptr(), scale2[1:2].data_ptr(), out4[2:3].data_ptr(), gb.data_ptr(), gs.data_ptr(), stream), 'backward')

        def fwd16_step():
            _lib.check(lib.odbox_match_loss_f16(*common, hb.data_ptr(), hs.data_ptr(), 1.5, 0.25, 1.0, out4.data_ptr(),
                                                None, None, ws.data_ptr(), ws.numel(), stream), 'forward_f16')

        def bwd16_step():
            _lib.check(lib.odbox_match_loss_backward_f16(*common, hb.data_ptr(), hs.data_ptr(), 1.5, 0.25, 1.0,
                                                         scale2[0:1].data_ptr(), scale2[1:2].data_ptr(), out4[2:3].data_ptr(), ghb.data_ptr(), ghs.data_ptr(), stream),
                       'backward_f16')
        fixed = B * (N * 16 + 20 * G)                                          # anchors + ground truth, always fp32
        for name, fn, byt in (('backward', bwd_step, fixed + 2 * B * N * (16 + 4 * C)),
                              ('forward_f16', fwd16_step, fixed + B * N * (8 + 2 * C)),
                              ('backward_f16', bwd16_step, fixed + 2 * B * N * (8 + 2 * C))):
            wl, el = timed(fn, args.loss_steps, 5, world)
            ach = byt / (el / args.loss_steps) / 1e9
            result['match_loss'][name] = {
                'value': round(world * B * args.loss_steps / wl, 1), 'unit': 'images/s',
                'us_per_step': round(el / args.loss_steps * 1e6, 2), 'steps': args.loss_steps,
                'roofline': {'bound': 'hbm', 'kernel': 'match_loss_kernel', 'achieved': round(ach, 1),
                             'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                             'bytes_per_launch': byt, 'traffic': None}}
        del pb, ps, gb, gs, hb, hs, ghb, ghs

    # ---------------- decode + NMS, configs[4] per-GPU share -------------------------------
    if not args.no_nms:
        S5, B5, C5 = 512, 64, 90
        anchors5 = A.generate_anchors(A.AnchorsConfig(), S5)
        N5 = anchors5.shape[0]
        reg5_np, sc5_np = workloads.nms_inputs(N5, C=C5, batch=B5, seed=1005 + 7919 * rank,
                                               p3p4_anchors=9 * (64 * 64 + 32 * 32))
        reg5 = torch.as_tensor(reg5_np).to(dev)
        sc5 = torch.as_tensor(sc5_np).to(dev)
        cb5 = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cb5 = cpu_baseline_nms(anchors5.cpu().numpy(), reg5_np, sc5_np, S5)
        del reg5_np, sc5_np
        out = {}

        def nms_step():
            b, s, l = BB.decode_and_nms(anchors5, reg5, sc5, (S5, S5))
            out['det'] = oddist.gather_detections(b, s, l, batch=None)

        wall5, ev5 = timed(nms_step, args.nms_steps, 3, world)
        bytes5 = B5 * workloads.nms_bytes_per_image(N5, C5)
        ach5 = bytes5 / (ev5 / args.nms_steps) / 1e9
        result['decode_nms'] = {
            'metric': 'images/sec decode+clip+per-class NMS+top-k' + (' + RCCL all-gather' if world > 1 else ''),
            'value': round(world * B5 * args.nms_steps / wall5, 1), 'unit': 'images/s',
            'ms_per_step': round(wall5 / args.nms_steps * 1e3, 4), 'steps': args.nms_steps,
            'config': {'workload': 'configs[4] per-GPU share: %d anchors, %d classes, batch %d per GPU, '
                                   'score 0.3, IoU 0.45, max_ret 100, ~4500 candidates/img' % (N5, C5, B5)},
            'roofline': {'bound': 'hbm', 'kernel': 'whole decode+NMS pipeline (score_scan dominates)',
                         'achieved': round(ach5, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(ach5 / HBM_PEAK_GBS, 4), 'bytes_per_launch': bytes5,
                         'traffic': load_traffic('score_scan_kernel')},
            'detections_per_image_mean': float((out['det'][1] > 0).sum().item()) / (world * B5),
        }
        if cb5 is not None:
            result['decode_nms']['cpu_baseline'] = {
                'value': round(cb5['multi'][0], 1), 'unit': 'images/s', 'cores': cb5['multi'][1], 'kind': 'port',
                'sample': 'oracle/odbox_oracle.c decode + per-class NMS + top-k on the first 16 images of the same '
                          'batch, repeated to %d images, OpenMP over images' % cb5['multi'][2],
                'value_1core': round(cb5['single'][0], 1)}

    # ---------------- CPU baseline (rank 0, N = 1 only) ------------------------------------
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


def load_traffic(kernel):
    """HBM bytes per launch from the PMC passes (profiles/pmc_traffic.json, written from the
    rocprofv3 --pmc runs of profiles/README.md); null until collected."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


if __name__ == '__main__':
    main()
