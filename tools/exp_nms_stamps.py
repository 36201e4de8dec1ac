#!/usr/bin/env python3
"""Where the tail kernels of decode+NMS spend their time: phase stamps of a -DODBOX_STAMPS build
(ODBOX_EXTRA_HIPCC_FLAGS=-DODBOX_STAMPS python -m object_detection_amd.build --force; ODBOX_LIBRARY=<that .so>).
Sums / maxima over every wave (or block) of ONE call on the configs[4] share; such a build drains memory at every
stamp, so the shares are what to read, not the lengths."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _lib, anchors as A, bndbox as BB, workloads  # noqa: E402

NAMES = {0: 'partition: count', 1: 'partition: keys + histogram', 2: 'partition: scan / starts / coop list', 3: 'partition: scatter',
         56: 'finalize: count', 57: 'finalize: keys to LDS', 58: 'finalize: sorts + merges', 59: 'finalize: emit'}
for base, what in ((8, 'wave seg m<=16'), (16, 'wave seg m<=64'), (24, 'wave seg m<=128')):
    for i, ph in enumerate(('bounds', 'keys + sort', 'box gather', 'walk', 'slot atomic')):
        NAMES[base + i] = '%s: %s' % (what, ph)
for base, what in ((32, 'coop <=3 chunks'), (40, 'coop 4 chunks'), (48, 'coop >4 chunks')):
    for i, ph in enumerate(('keys + rank sort', 'box gather', 'chunk loop', 'slot atomic')):
        NAMES[base + i] = '%s: %s' % (what, ph)


# (the fused pass-1 block, tier_tail_kernel, uses slots 40 .. 49; the cooperative blocks do not run beside it)
FUSED = {40: 'fused: count', 41: 'fused: keys loaded', 42: 'fused: tier selection', 43: 'fused: class histogram / starts / scatter',
         44: 'fused: rank sort', 45: 'fused: box gather + publish', 46: 'fused: rounds (per wave)', 47: 'fused: final keys to LDS (+ slowest wave)',
         48: 'fused: chunk sorts + merges', 49: 'fused: emit'}


def main():
    lib = _lib.load()
    dev = _lib.require_gpu()
    if os.environ.get('ODBOX_STAMPS_FUSED', '1') != '0':      # ODBOX_STAMPS_FUSED=0: the separate kernels
        NAMES.update(FUSED)
    else:
        BB.set_nms_fused_tail(False)
    if os.environ.get('ODBOX_STAMPS_TIER'):                   # ODBOX_STAMPS_TIER=0: no score tier, the single plain pass (pass 2's kernels)
        BB.set_nms_tier(int(os.environ['ODBOX_STAMPS_TIER']))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    c = workloads.CONFIGS[5]
    anchors = A.generate_anchors(A.AnchorsConfig(), c['S'])
    N = anchors.shape[0]
    reg, sc = workloads.nms_inputs(N, C=c['C'], batch=B, p3p4_anchors=9 * (64 * 64 + 32 * 32))
    reg, sc = torch.as_tensor(reg).to(dev), torch.as_tensor(sc).to(dev)
    arr = (ctypes.c_ulonglong * 64)
    s, m, n = arr(), arr(), arr()
    for _ in range(3):
        BB.decode_and_nms(anchors, reg, sc, (c['S'], c['S']), pre_nms_topk=5000)
    lib.odbox_debug_stamps(s, m, n, 1)
    BB.decode_and_nms(anchors, reg, sc, (c['S'], c['S']), pre_nms_topk=5000)
    lib.odbox_debug_stamps(s, m, n, 1)
    print('B = %d; per phase: waves (or blocks), mean us, max us' % B)
    for i in range(64):
        if n[i]:
            print('%-42s %6d  %7.2f  %7.2f' % (NAMES.get(i, 'slot %d' % i), n[i], s[i] / n[i] / 100.0, m[i] / 100.0))


if __name__ == '__main__':
    main()
