"""decode + NMS latency for small batches (serving: infdet.py feeds a few images at a time)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import anchors as A, bndbox as BB, workloads  # noqa: E402

S, C = 512, 90
if len(sys.argv) > 1:                       # 0: the separate pass-1 kernels, 1: the fused block (bndbox.set_nms_fused_tail)
    BB.set_nms_fused_tail(bool(int(sys.argv[1])))
    print('fused tail:', sys.argv[1])
anchors = A.generate_anchors(A.AnchorsConfig(), S)
N = anchors.shape[0]
for B in (1, 2, 4, 8, 16, 32, 64):
    reg, sc = workloads.nms_inputs(N, C=C, batch=B, seed=5, p3p4_anchors=9 * (64 * 64 + 32 * 32))
    reg, sc = torch.as_tensor(reg).cuda(), torch.as_tensor(sc).cuda()
    for _ in range(5):
        BB.decode_and_nms(anchors, reg, sc, (S, S))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = BB.decode_and_nms(anchors, reg, sc, (S, S))
    res = {}
    for name, fn in (('eager', lambda: BB.decode_and_nms(anchors, reg, sc, (S, S))), ('graph', g.replay)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + '_us'] = round(e0.elapsed_time(e1) * 20, 1)
    print(json.dumps(dict(B=B, **res, img_s_graph=round(B / (res['graph_us'] * 1e-6)))))
