#!/usr/bin/env python3
"""decode+NMS on a batch shaped like a TRAINED detector's output rather than BASELINE's synthetic one: every object
draws a pile of anchors whose regressors decode to nearly the object's own box (so the piles collapse under NMS: one
candidate in dozens survives and the whole list holds fewer than max_ret detections), 40 objects x ~100 anchors per
image on the configs[4] grid.  Times one call after the other with the tier ladder (default), with the separate
pass-1 kernels and with the tier off, and prints which rung finished the images.

    python tools/exp_nms_overlap.py [objects=40] [pile=100] [noise=0.03]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _lib, anchors as A, bndbox as BB  # noqa: E402


def make(anchors, B, C, objects, pile, noise, rng):
    N = anchors.shape[0]
    reg = np.clip(rng.normal(0, 0.5, (B, N, 4)), -2, 2).astype(np.float32)
    sc = rng.uniform(0, 0.25, (B, N, C)).astype(np.float32)
    ctr = (anchors[:, :2] + anchors[:, 2:]) / 2
    wh = anchors[:, 2:] - anchors[:, :2]
    for b in range(B):
        for _ in range(objects):
            size = np.exp(rng.uniform(np.log(24), np.log(200), 2))
            c = rng.uniform(60, 452, 2)
            box = np.array([c[0] - size[0] / 2, c[1] - size[1] / 2, c[0] + size[0] / 2, c[1] + size[1] / 2], np.float32)
            # the anchors a matcher would call positive for this box: nearest centres among anchors of similar size
            d = np.abs(ctr - c).sum(1) / np.sqrt(size.prod()) + np.abs(np.log(wh.prod(1) / size.prod()))
            idx = np.argpartition(d, pile)[:pile]
            a = anchors[idx]
            aw, ah = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
            target = box + rng.normal(0, noise, (pile, 4)).astype(np.float32) * np.tile(size, 2).astype(np.float32)
            reg[b, idx] = np.stack([(target[:, 0] - a[:, 0]) / aw, (target[:, 1] - a[:, 1]) / ah,
                                    (target[:, 2] - a[:, 2]) / aw, (target[:, 3] - a[:, 3]) / ah], 1) / 0.2
            sc[b, idx, int(rng.integers(0, C))] = (0.3 + 0.7 * rng.uniform(size=pile)).astype(np.float32)
    return reg, sc


def main():
    objects = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    pile = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
    lib = _lib.load()
    dev = _lib.require_gpu()
    S, B, C = 512, 64, 90
    anchors = A.generate_anchors(A.AnchorsConfig(), S)
    an = anchors.cpu().numpy()
    N = an.shape[0]
    rng = np.random.default_rng(7)
    reg8, sc8 = make(an, 8, C, objects, pile, noise, rng)
    reg = torch.as_tensor(np.tile(reg8, (8, 1, 1))).to(dev)
    sc = torch.as_tensor(np.tile(sc8, (8, 1, 1))).to(dev)
    res = {}
    for name, fused, tier in (('ladder (default)', True, 8), ('separate pass-1 kernels', False, 8), ('tier off', True, 0)):
        BB.set_nms_fused_tail(fused)
        BB.set_nms_tier(tier)
        for _ in range(5):
            out = BB.decode_and_nms(anchors, reg, sc, (S, S), return_keep=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                BB.decode_and_nms(anchors, reg, sc, (S, S))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 50)
        ws = _lib.workspace('decode_nms', dev, lib.odbox_decode_nms_workspace(B, N, C, 100, 0))
        state = ws[256 + 2 * B * 64 * 4:][:4 * B].view(torch.int32).cpu().tolist()
        res[name] = out
        print('%-26s %.1f us per call; detections per image %.1f; rungs %s' % (
            name, sorted(ts)[3], float(out[4].float().mean()), {k: state.count(k) for k in sorted(set(state))} if fused and tier else '-'))
    BB.set_nms_fused_tail(True)
    BB.set_nms_tier(8)
    ref = res['tier off']
    print('same detections:', all(all(torch.equal(a, b) for a, b in zip(v, ref)) for v in res.values()))


if __name__ == '__main__':
    main()
