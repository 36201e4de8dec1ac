#!/usr/bin/env python3
"""decode+NMS call after call, eager against replayed from a HIP graph (one call captured), for a small and the full
batch of the configs[4] share: what a serving loop gains by capturing the three launches.

    python tools/exp_nms_graph.py [batch ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _lib, anchors as A, workloads  # noqa: E402


def main():
    batches = [int(x) for x in sys.argv[1:]] or [1, 8, 64]
    lib = _lib.load()
    dev = _lib.require_gpu()
    c = workloads.CONFIGS[5]
    anchors = A.generate_anchors(A.AnchorsConfig(), c['S'])
    N, C, S = anchors.shape[0], c['C'], c['S']
    for B in batches:
        reg, sc = workloads.nms_inputs(N, C=C, batch=B, p3p4_anchors=9 * (64 * 64 + 32 * 32), distinct=min(B, 8))
        reg, sc = torch.as_tensor(reg).to(dev), torch.as_tensor(sc).to(dev)
        det_b = torch.empty((B, 100, 4), device=dev)
        det_s = torch.empty((B, 100), device=dev)
        det_l = torch.empty((B, 100), dtype=torch.int32, device=dev)
        det_k = torch.empty((B, 100), dtype=torch.int32, device=dev)
        det_n = torch.empty((B,), dtype=torch.int32, device=dev)
        ws = torch.empty(lib.odbox_decode_nms_workspace(B, N, C, 100, 5000), dtype=torch.uint8, device=dev)

        def enqueue():
            rc = lib.odbox_decode_nms(anchors.data_ptr(), reg.data_ptr(), sc.data_ptr(), B, N, C, float(S), float(S), 1,
                                      0.3, 0.45, 100, 5000, det_b.data_ptr(), det_s.data_ptr(), det_l.data_ptr(),
                                      det_k.data_ptr(), det_n.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            assert rc == 0, lib.odbox_last_error()

        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(5):
                enqueue()
            side.synchronize()
            eager_out = det_s.clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                enqueue()

            def timed(fn, iters=50, groups=5):
                ts = []
                for _ in range(groups):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        fn()
                    e1.record()
                    side.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / iters)
                return sorted(ts)[len(ts) // 2]
            t_eager = timed(enqueue)
            t_graph = timed(graph.replay)
            assert torch.equal(det_s, eager_out)
        print('batch %3d: eager %.1f us per call, graph replay %.1f us per call' % (B, t_eager, t_graph))


if __name__ == '__main__':
    main()
