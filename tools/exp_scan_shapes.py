"""decode+NMS across class counts and anchor counts: is there a shape where the score scan falls off its
streaming rate?  Almost no candidates (scores below the threshold except 50 per image), so the call is the
scan plus the fixed cost of the tail kernels.  Prints us per call and the GB/s of the score tensor alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import anchors as A, bndbox as BB   # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    anchors_full = A.generate_anchors(A.AnchorsConfig(), 512).to(dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    for N in (49104, 10647):
        anchors = anchors_full[:N].contiguous()
        for C in (1, 2, 3, 4, 8, 20, 21, 80, 81, 90, 91, 128, 256, 365):
            B = max(1, min(64, int(1.2e9 // (N * C * 4))))
            sc = torch.rand((B, N, C), device=dev, generator=gen) * 0.25
            idx = torch.randint(0, N, (B, 50), device=dev, generator=gen)
            sc[torch.arange(B, device=dev)[:, None], idx, 0] = 0.9
            reg = torch.zeros((B, N, 4), device=dev)
            for _ in range(3):
                BB.decode_and_nms(anchors, reg, sc, (512, 512))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                BB.decode_and_nms(anchors, reg, sc, (512, 512))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100.0
            byts = B * N * C * 4
            print('N %6d C %4d B %3d  %8.1f us per call  scores %7.1f MB  %6.0f GB/s' % (N, C, B, us, byts / 1e6, byts / us / 1e3), flush=True)
            del sc, reg


if __name__ == '__main__':
    main()
