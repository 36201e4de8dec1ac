#!/usr/bin/env python3
"""One decode+NMS call after the other on the configs[4] share (49 104 anchors, 90 classes, 64 images, pre-NMS top-k
5 000): HIP-event time per call, median over groups, and a digest of the detections (two builds that print the same
digest returned the same bits).  The A/B aid for everything behind the score scan.

    python tools/exp_nms_single.py [--iters 20] [--groups 7] [--batch 64] [--tier F] [--fused 0|1]      (F: bndbox.set_nms_tier, 0 = off)
"""
import argparse
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _lib, anchors as A, bndbox as BB, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--groups', type=int, default=7)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--tier', type=int, default=None)
    ap.add_argument('--distinct', type=int, default=8, help='distinct synthetic images in the batch (bench.py: the whole batch)')
    ap.add_argument('--fused', type=int, default=None, help='bndbox.set_nms_fused_tail: 1 = pass 1 as one launch (default), 0 = separate kernels')
    args = ap.parse_args()
    _lib.load()
    if args.tier is not None:
        BB.set_nms_tier(args.tier)
    if args.fused is not None:
        BB.set_nms_fused_tail(bool(args.fused))
    dev = _lib.require_gpu()
    c = workloads.CONFIGS[5]
    anchors = A.generate_anchors(A.AnchorsConfig(), c['S'])
    N = anchors.shape[0]
    reg, sc = workloads.nms_inputs(N, C=c['C'], batch=args.batch, p3p4_anchors=9 * (64 * 64 + 32 * 32), distinct=args.distinct)
    reg, sc = torch.as_tensor(reg).to(dev), torch.as_tensor(sc).to(dev)

    def call():
        return BB.decode_and_nms(anchors, reg, sc, (c['S'], c['S']), pre_nms_topk=5000, return_keep=True)
    for _ in range(5):
        out = call()
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for t in out:
        h.update(t.cpu().numpy().tobytes())
    times = []
    for _ in range(args.groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / args.iters)
    lib = _lib.load()
    ws = _lib.workspace('decode_nms', dev, lib.odbox_decode_nms_workspace(args.batch, N, c['C'], 100, 5000))
    state = ws[256 + 2 * args.batch * 64 * 4:][:4 * args.batch].view(torch.int32).cpu().tolist()
    print('images by ladder rung (0 = pass 2; meaningful with the fused block only):', {k: state.count(k) for k in sorted(set(state))})
    times.sort()
    byts = args.batch * workloads.nms_bytes_per_image(N, c['C'])
    med = times[len(times) // 2]
    print('%s tier %s fused %s: per call min %.1f median %.1f max %.1f us  -> %.3f of 8 TB/s   digest %s' % (
        os.environ.get('ODBOX_LIBRARY', 'default').split('libodbox_')[-1], args.tier, args.fused, times[0], med, times[-1], byts / med / 8e6, h.hexdigest()[:12]))


if __name__ == '__main__':
    main()
