"""match+encode on the config-2 grid for a sweep of batch sizes (dense output 55 ... 350 MB): where
does the best store policy for the zero stream change?  Run under library variants (ODBOX_LIBRARY)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _lib, workloads  # noqa: E402
from oracle import odbox_oracle as O  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda')
S, C = 416, 80
anchors = torch.as_tensor(O.yolo_shaped_anchors(S)).to(dev)
N = anchors.shape[0]
st = torch.cuda.current_stream().cuda_stream
row = {}
for B in [int(b) for b in os.environ.get('SWEEP_B', '16 32 40 48 56 64 80 100').split()]:
    gt, lab = workloads.match_inputs(2, batch=B)
    g, l = torch.as_tensor(gt).to(dev), torch.as_tensor(lab).to(dev)
    reg = torch.empty((B, N, 5), device=dev)
    labels = torch.empty((B, N, C + 1), device=dev)

    def step():
        assert lib.odbox_match_encode(anchors.data_ptr(), g.data_ptr(), l.data_ptr(), B, N, g.shape[1], C, float(S), float(S),
                                      0.4, 0.5, reg.data_ptr(), labels.data_ptr(), None, None, None, None, st) == 0
    for _ in range(10):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        step()
    e1.record()
    torch.cuda.synchronize()
    row['B%d/%dMB' % (B, B * N * (C + 1) * 4 // 1000000)] = round(e0.elapsed_time(e1) * 20, 1)
print(json.dumps(row))
