"""Latency of the NMS kernel on ONE (image, class) segment: the longest segment of image 0 of the
cfg-5 workload alone, then the whole image, then 64 images (run under rocprofv3 --kernel-trace)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import anchors as A, bndbox as BB, workloads

anchors = A.generate_anchors(A.AnchorsConfig(), 512)
N = anchors.shape[0]
reg, sc = workloads.nms_inputs(N, C=90, batch=8, seed=1005, p3p4_anchors=9 * (64 * 64 + 32 * 32))
best, lab = sc[0].max(-1), sc[0].argmax(-1)
cnt = np.bincount(lab[best > 0.3], minlength=90)
c = int(cnt.argmax())
one = sc[:1].copy()
mask = (lab == c) & (best > 0.3)
one[0, ~mask] = 0
print('longest segment: class', c, 'm =', int(cnt[c]))
for name, r, s in (('one_segment', reg[:1], one), ('one_image', reg[:1], sc[:1]), ('eight_images', reg, sc)):
    r, s = torch.as_tensor(r).cuda(), torch.as_tensor(s).cuda()
    for _ in range(3):
        out = BB.decode_and_nms(anchors, r, s, (512, 512), return_keep=True)
    torch.cuda.synchronize()
    print(name, 'count', out[4].tolist()[:2])
