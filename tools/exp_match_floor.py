"""Where does match+encode's time go on the bench shape?  Same launch with 20, 1 far-away and 1
all-padded GT row: the store stream is identical, only the match phase shrinks."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _lib, workloads  # noqa: E402
from oracle import odbox_oracle as O  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda')
S, B, C = 416, 64, 80
anchors = torch.as_tensor(O.yolo_shaped_anchors(S)).to(dev)
N = anchors.shape[0]
gt20, lab20 = workloads.match_inputs(2)
cases = {'G20': (gt20, lab20),
         'G1_far': (np.full((B, 1, 4), 5000, np.float32) + np.float32([0, 0, 10, 10]), np.zeros((B, 1), np.int32)),
         'G1_pad': (np.full((B, 1, 4), -1, np.float32), np.full((B, 1), -1, np.int32))}
reg = torch.empty((B, N, 5), device=dev)
labels = torch.empty((B, N, C + 1), device=dev)
st = torch.cuda.current_stream().cuda_stream
for name, (gt, lab) in cases.items():
    g, l = torch.as_tensor(gt).to(dev), torch.as_tensor(lab).to(dev)
    G = g.shape[1]

    def step():
        assert lib.odbox_match_encode(anchors.data_ptr(), g.data_ptr(), l.data_ptr(), B, N, G, C, float(S), float(S), 0.4, 0.5,
                                      reg.data_ptr(), labels.data_ptr(), None, None, None, None, st) == 0
    for _ in range(10):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({'case': name, 'us': e0.elapsed_time(e1) * 10}))
x = torch.empty(B * N * (C + 1 + 5), device=dev)
for _ in range(10):
    x.fill_(1.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(100):
    x.fill_(1.0)
e1.record()
torch.cuda.synchronize()
print(json.dumps({'case': 'torch fill of the same bytes', 'us': e0.elapsed_time(e1) * 10}))
