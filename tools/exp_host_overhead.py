"""Host-side cost of one training-step loss through the Python mirror (cProfile, config 2)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import metric as M, workloads  # noqa: E402
from oracle import odbox_oracle as O  # noqa: E402

dev = torch.device('cuda')
S, B, C = 416, 64, 80
anchors = torch.as_tensor(O.yolo_shaped_anchors(S)).to(dev)
N = anchors.shape[0]
gt, lab = workloads.match_inputs(2, batch=B)
gt, lab = torch.as_tensor(gt).to(dev), torch.as_tensor(lab).to(dev)
pb = torch.randn((B, N, 4), device=dev, requires_grad=True)
ps = torch.rand((B, N, C), device=dev, requires_grad=True)


def step():
    pb.grad = None
    ps.grad = None
    d, c, _ = M.detection_loss(anchors, (S, S), gt, lab, pb, ps, C)
    (d + c).backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host issue time per step %.1f us, with final sync %.1f us' % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
