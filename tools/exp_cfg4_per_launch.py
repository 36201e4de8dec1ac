#!/usr/bin/env python3
"""configs[3] (500 GT rows, VALU-bound) launch by launch: the HIP-event time of every one of N consecutive
odbox_match_encode launches after an idle second -- shows whether the clock the launches run at sags under the
sustained VALU load (the SQ cycle counters of a launch divided by its duration give the clock).

    python tools/exp_cfg4_per_launch.py [launches]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _lib, anchors as A, workloads  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lib = _lib.load()
    dev = _lib.require_gpu()
    st = _lib.stream_ptr()
    c = workloads.CONFIGS[4]
    anchors = A.generate_anchors(A.AnchorsConfig(), c['S'])
    N, B = anchors.shape[0], 128
    gt, lab = workloads.match_inputs(4, batch=B)
    gt, lab = torch.as_tensor(gt).to(dev), torch.as_tensor(lab).to(dev)
    G, C = gt.shape[1], c['C']
    reg = torch.empty((B, N, 5), device=dev)
    out = torch.empty((B, N, C + 1), device=dev)

    def step():
        rc = lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(c['S']),
                                    float(c['S']), 0.4, 0.5, reg.data_ptr(), out.data_ptr(), None, None, None, None, st)
        assert rc == 0
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for rep in range(2):
        time.sleep(1.0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
        print(os.environ.get('ODBOX_LIBRARY', 'default').split('libodbox_')[-1], 'rep', rep, ' '.join('%.0f' % u for u in us))


if __name__ == '__main__':
    main()
