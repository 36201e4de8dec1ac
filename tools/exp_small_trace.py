"""decode + NMS on ONE image, 200 calls: per-kernel durations under rocprofv3 --kernel-trace --stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import anchors as A, bndbox as BB, workloads
S, C = 512, 90
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
anchors = A.generate_anchors(A.AnchorsConfig(), S)
N = anchors.shape[0]
reg, sc = workloads.nms_inputs(N, C=C, batch=B, seed=5, p3p4_anchors=9 * (64 * 64 + 32 * 32))
reg, sc = torch.as_tensor(reg).cuda(), torch.as_tensor(sc).cuda()
for _ in range(200):
    BB.decode_and_nms(anchors, reg, sc, (S, S))
torch.cuda.synchronize()
