#!/usr/bin/env python3
"""All five BASELINE.json configurations on ONE MI355X (per-GPU shares for the multi-GPU ones):
HIP-event time per launch, images/s, fraction of the roofline that bounds it.  Documentation
aid (BASELINE.md / DESIGN.md tables); the driver's bench is bench.py.

    python tools/bench_configs.py [--cfg 1 2 3 4 5] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _lib, anchors as A, bndbox as BB, workloads  # noqa: E402

HBM = 8000.0


def event_time(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def grid(cfg):
    c = workloads.CONFIGS[cfg]
    if c['grid'] == 'yolo':
        parts = [A.AnchorGenerator(s, [1], st)((c['S'] // st, c['S'] // st, 3)) for s, st in ((32, 8), (64, 16), (128, 32))]
        return torch.cat(parts, 0)
    return A.generate_anchors(A.AnchorsConfig(), c['S'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', type=int, nargs='*', default=[1, 2, 3, 4, 5, 6, 7],
                    help='1-5: BASELINE.json configs; 6: evaluation side (COCO matching) on the gathered cfg-5 batch; '
                         '7: ground-truth preparation of a 64-image batch')
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    lib = _lib.load()
    dev = _lib.require_gpu()
    st = _lib.stream_ptr()
    rows = []
    for cfg in [x for x in args.cfg if x <= 5]:
        c = workloads.CONFIGS[cfg]
        anchors = grid(cfg)
        N = anchors.shape[0]
        if cfg == 5:
            B = 64
            reg, sc = workloads.nms_inputs(N, C=c['C'], batch=B, p3p4_anchors=9 * (64 * 64 + 32 * 32))
            reg, sc = torch.as_tensor(reg).to(dev), torch.as_tensor(sc).to(dev)
            # BASELINE names a pre-NMS top-k of 5000 (it never binds on these inputs); measured in turns with the unbounded call
            t0 = event_time(lambda: BB.decode_and_nms(anchors, reg, sc, (c['S'], c['S'])), args.iters)
            t = event_time(lambda: BB.decode_and_nms(anchors, reg, sc, (c['S'], c['S']), pre_nms_topk=5000), args.iters)
            t0 = min(t0, event_time(lambda: BB.decode_and_nms(anchors, reg, sc, (c['S'], c['S'])), args.iters))
            t = min(t, event_time(lambda: BB.decode_and_nms(anchors, reg, sc, (c['S'], c['S']), pre_nms_topk=5000), args.iters))
            byts = B * workloads.nms_bytes_per_image(N, c['C'])
            rows.append(dict(cfg=cfg, B=B, N=N, us=t * 1e6, img_s=B / t, GBs=byts / t / 1e9, frac_hbm=byts / t / 1e9 / HBM,
                             pre_nms_topk=5000, us_without_topk=t0 * 1e6))
            # 16-bit model outputs (odbox_decode_nms_mixed): half the score bytes, same detections as the widened tensors
            for name, dt in (('5_bf16_scores', torch.bfloat16), ('5_f16_scores', torch.float16)):
                sc16, reg16 = sc.to(dt), reg.to(dt)
                t16 = min(event_time(lambda: BB.decode_and_nms(anchors, reg16, sc16, (c['S'], c['S']), pre_nms_topk=5000), args.iters)
                          for _ in range(2))
                b16 = B * (N * (16 + 8 + 2 * c['C']) + 24 * 100)
                rows.append(dict(cfg=name, B=B, N=N, us=t16 * 1e6, img_s=B / t16, GBs=b16 / t16 / 1e9, frac_hbm=b16 / t16 / 1e9 / HBM))
                del sc16, reg16
            continue
        B = {1: 8, 2: 64, 3: 32, 4: 128}[cfg]
        gt, lab = workloads.match_inputs(cfg, batch=B)
        gt, lab = torch.as_tensor(gt).to(dev), torch.as_tensor(lab).to(dev)
        G, C = gt.shape[1], c['C']
        reg = torch.empty((B, N, 5), device=dev)
        out = torch.empty((B, N, C + 1), device=dev)

        def step():
            rc = lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(c['S']),
                                        float(c['S']), 0.4, 0.5, reg.data_ptr(), out.data_ptr(), None, None, None, None, st)
            assert rc == 0
        t = event_time(step, args.iters)
        byts = B * workloads.match_bytes_per_image(N, G, C)
        row = dict(cfg=cfg, B=B, N=N, G=G, C=C, us=t * 1e6, img_s=B / t, GBs=byts / t / 1e9, frac_hbm=byts / t / 1e9 / HBM)
        if cfg == 4:
            # VALU-bound.  No "pairs / peak rate" fraction: the kernel proves most pairs zero without evaluating
            # them, so that ratio exceeds 1; the SQ counters (tools/pmc_cfg.sh 4) say how busy the VALUs are
            row['bound'] = 'valu (see profiles/pmc_traffic.json: match_encode_cfg3_valu_busy)'
        rows.append(row)
        if cfg == 2:
            # the optional compact-output mode (match index, state, 4 offsets + state column =
            # 25 B per anchor instead of 344), reported separately: what the matcher costs without the dense labels
            idx = torch.empty((B, N), dtype=torch.int32, device=dev)
            state = torch.empty((B, N), dtype=torch.int8, device=dev)

            def step_compact():
                rc = lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(c['S']),
                                            float(c['S']), 0.4, 0.5, reg.data_ptr(), None, idx.data_ptr(), state.data_ptr(),
                                            None, None, st)
                assert rc == 0
            t = event_time(step_compact, args.iters)
            byts = B * (N * (16 + 20 + 5) + 20 * G)
            rows.append(dict(cfg='2_compact', B=B, N=N, G=G, C=C, us=t * 1e6, img_s=B / t, GBs=byts / t / 1e9,
                             frac_hbm=byts / t / 1e9 / HBM))
    # fused matcher + loss on the cfg-2 batch (forward, and forward + gradients)
    if 2 in args.cfg:
        c = workloads.CONFIGS[2]
        anchors = grid(2)
        N, B, C = anchors.shape[0], 64, c['C']
        gt, lab = workloads.match_inputs(2, batch=B)
        gt, lab = torch.as_tensor(gt).to(dev), torch.as_tensor(lab).to(dev)
        G = gt.shape[1]
        pb = torch.randn((B, N, 4), device=dev)
        ps = torch.rand((B, N, C), device=dev)
        out = torch.empty(4, device=dev)
        gb, gs = torch.empty_like(pb), torch.empty_like(ps)
        ws = torch.empty(lib.odbox_match_loss_workspace(B, N), dtype=torch.uint8, device=dev)
        for name, g1, g2 in (('2_loss_fwd', None, None), ('2_loss_fwd_bwd', gb, gs)):
            def step():
                rc = lib.odbox_match_loss(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(c['S']),
                                          float(c['S']), 0.4, 0.5, pb.data_ptr(), ps.data_ptr(), 1.5, 0.25, 1.0,
                                          out.data_ptr(), None if g1 is None else g1.data_ptr(),
                                          None if g2 is None else g2.data_ptr(), ws.data_ptr(), ws.numel(), st)
                assert rc == 0
            t = event_time(step, args.iters)
            byts = B * (N * (32 + 4 * C) + 20 * G) * (1 if g1 is None else 2)
            rows.append(dict(cfg=name, B=B, N=N, G=G, C=C, us=t * 1e6, img_s=B / t, GBs=byts / t / 1e9,
                             frac_hbm=byts / t / 1e9 / HBM))
        # the autograd backward launch: gradients only, times two device-resident factors
        scale = torch.tensor([0.5, 0.25], device=dev)

        def step_bwd():
            rc = lib.odbox_match_loss_backward(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(c['S']),
                                               float(c['S']), 0.4, 0.5, pb.data_ptr(), ps.data_ptr(), 1.5, 0.25, 1.0,
                                               scale[0:1].data_ptr(), scale[1:2].data_ptr(), out[2:3].data_ptr(), gb.data_ptr(), gs.data_ptr(), st)
            assert rc == 0
        t = event_time(step_bwd, args.iters)
        byts = B * (N * (32 + 4 * C) + 20 * G) * 2
        rows.append(dict(cfg='2_loss_backward', B=B, N=N, G=G, C=C, us=t * 1e6, img_s=B / t, GBs=byts / t / 1e9,
                         frac_hbm=byts / t / 1e9 / HBM))
        # fp16 predictions and gradients (the reference's --use_fp16 run): half the stream
        hb, hs = pb.half(), ps.half()
        ghb, ghs = torch.empty_like(hb), torch.empty_like(hs)

        def step_h_fwd():
            rc = lib.odbox_match_loss_f16(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(c['S']),
                                          float(c['S']), 0.4, 0.5, hb.data_ptr(), hs.data_ptr(), 1.5, 0.25, 1.0,
                                          out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st)
            assert rc == 0

        def step_h_bwd():
            rc = lib.odbox_match_loss_backward_f16(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C,
                                                   float(c['S']), float(c['S']), 0.4, 0.5, hb.data_ptr(), hs.data_ptr(),
                                                   1.5, 0.25, 1.0, scale[0:1].data_ptr(), scale[1:2].data_ptr(), out[2:3].data_ptr(), ghb.data_ptr(), ghs.data_ptr(), st)
            assert rc == 0
        bb, bs = pb.bfloat16(), ps.bfloat16()
        gbb, gbs = torch.empty_like(bb), torch.empty_like(bs)

        def step_b_fwd():
            rc = lib.odbox_match_loss_bf16(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(c['S']),
                                           float(c['S']), 0.4, 0.5, bb.data_ptr(), bs.data_ptr(), 1.5, 0.25, 1.0,
                                           out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st)
            assert rc == 0

        def step_b_bwd():
            rc = lib.odbox_match_loss_backward_bf16(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C,
                                                    float(c['S']), float(c['S']), 0.4, 0.5, bb.data_ptr(), bs.data_ptr(),
                                                    1.5, 0.25, 1.0, scale[0:1].data_ptr(), scale[1:2].data_ptr(),
                                                    out[2:3].data_ptr(), gbb.data_ptr(), gbs.data_ptr(), st)
            assert rc == 0
        for name, fn, mult in (('2_loss_fwd_f16', step_h_fwd, 1), ('2_loss_backward_f16', step_h_bwd, 2),
                               ('2_loss_fwd_bf16', step_b_fwd, 1), ('2_loss_backward_bf16', step_b_bwd, 2)):
            t = event_time(fn, args.iters)
            byts = B * (N * 16 + mult * N * (8 + 2 * C) + 20 * G)
            rows.append(dict(cfg=name, B=B, N=N, G=G, C=C, us=t * 1e6, img_s=B / t, GBs=byts / t / 1e9,
                             frac_hbm=byts / t / 1e9 / HBM))
        # what a training step pays through the Python mirror: forward, then autograd backward of
        # dist + class (two HIP launches + loss_reduce; everything else is host-side bookkeeping)
        from object_detection_amd import metric as M
        tpb, tps = pb.clone().requires_grad_(True), ps.clone().requires_grad_(True)

        def step_autograd():
            tpb.grad = None
            tps.grad = None
            d, cl, _ = M.detection_loss(anchors, (c['S'], c['S']), gt, lab, tpb, tps, C)
            (d + cl).backward()
        t = event_time(step_autograd, args.iters)
        rows.append(dict(cfg='2_loss_autograd_step', B=B, N=N, G=G, C=C, us=t * 1e6, img_s=B / t))
    # COCO matching of the gathered detections of one cfg-5 eval batch (512 images)
    if 6 in args.cfg:
        import ctypes
        import time
        import numpy as np
        from object_detection_amd import cocoeval as CE
        B, R, G, C = 512, 100, 64, 90
        d = workloads.eval_inputs(B=B, R=R, G=G, C=C)
        dev_d = {k: torch.as_tensor(v).to(dev) for k, v in d.items()}
        p = CE.Params()
        thr = np.ascontiguousarray(p.iouThrs, dtype=np.float64)
        rng = np.ascontiguousarray(p.areaRng, dtype=np.float64)
        T, A = len(thr), len(rng)
        dtm = torch.empty((B, A, T, R), dtype=torch.int32, device=dev)
        dti = torch.empty((B, A, T, R), dtype=torch.uint8, device=dev)
        gtm = torch.empty((B, A, T, G), dtype=torch.int32, device=dev)
        gti = torch.empty((B, A, G), dtype=torch.uint8, device=dev)
        rank = torch.empty((B, R), dtype=torch.int32, device=dev)
        ws = torch.empty(lib.odbox_coco_match_workspace(B, R), dtype=torch.uint8, device=dev)

        def step():
            rc = lib.odbox_coco_match(dev_d['dt_boxes'].data_ptr(), dev_d['dt_scores'].data_ptr(),
                                      dev_d['dt_labels'].data_ptr(), dev_d['gt_boxes'].data_ptr(),
                                      dev_d['gt_labels'].data_ptr(), B, R, G, C, thr.ctypes.data_as(ctypes.c_void_p), T,
                                      rng.ctypes.data_as(ctypes.c_void_p), A, 100, dtm.data_ptr(), dti.data_ptr(),
                                      gtm.data_ptr(), gti.data_ptr(), rank.data_ptr(), ws.data_ptr(), ws.numel(), st)
            assert rc == 0
        t = event_time(step, args.iters)
        byts = B * (R * 24 + G * 20 + A * T * (R * 5 + G * 4) + A * G + R * 4)
        row = dict(cfg='6_coco_match', B=B, R=R, G=G, C=C, us=t * 1e6, img_s=B / t, GBs=byts / t / 1e9,
                   frac_hbm=byts / t / 1e9 / HBM, pairs_per_s=B * A * T * R * 20 / t)
        from oracle import coco_oracle as CO
        n = 16
        t0 = time.perf_counter()
        CO.dense_eval(d['dt_boxes'][:n], d['dt_scores'][:n], d['dt_labels'][:n], d['gt_boxes'][:n], d['gt_labels'][:n],
                      (512, 512), C)
        row['cpu_oracle_img_s'] = n / (time.perf_counter() - t0)
        rows.append(row)
    # ground-truth preparation, ragged annotations of 64 images -> [64,64,4] / [64,64]
    if 7 in args.cfg:
        import time
        import numpy as np
        from oracle import odbox_oracle as OO
        rng = np.random.default_rng(1007)
        B, max_items, S = 64, 64, 512
        counts = rng.integers(5, 60, B)
        hw = np.stack([rng.integers(300, 1200, B), rng.integers(300, 1200, B)], 1).astype(np.float32)
        boxes = [np.stack([rng.uniform(0, 200, n), rng.uniform(0, 200, n), rng.uniform(8, 90, n), rng.uniform(8, 90, n)], 1).astype(np.float32) for n in counts]
        labels = [rng.integers(0, 80, n).astype(np.int32) for n in counts]
        offs = np.zeros(B + 1, np.int32)
        np.cumsum(counts, out=offs[1:])
        d_boxes = torch.as_tensor(np.concatenate(boxes)).to(dev)
        d_lab = torch.as_tensor(np.concatenate(labels)).to(dev)
        d_offs = torch.as_tensor(offs).to(dev)
        d_hw = torch.as_tensor(hw).to(dev)
        ob = torch.empty((B, max_items, 4), device=dev)
        ol = torch.empty((B, max_items), dtype=torch.int32, device=dev)

        def step():
            rc = lib.odbox_gt_prepare_xywh(d_boxes.data_ptr(), d_lab.data_ptr(), d_offs.data_ptr(), int(d_lab.shape[0]), d_hw.data_ptr(), B, max_items,
                                           float(S), 0, 0, ob.data_ptr(), ol.data_ptr(), st)
            assert rc == 0
        t = event_time(step, args.iters)
        t0 = time.perf_counter()
        for _ in range(20):
            OO.padded_batch_boxes([OO.prepare_example_boxes(bb, ll, h, S, max_items) for bb, ll, h in zip(boxes, labels, hw)], max_items)
        cpu = B * 20 / (time.perf_counter() - t0)
        rows.append(dict(cfg='7_gt_prepare', B=B, rows=int(offs[-1]), us=t * 1e6, img_s=B / t, cpu_oracle_img_s=cpu))
    for r in rows:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))


if __name__ == '__main__':
    main()
