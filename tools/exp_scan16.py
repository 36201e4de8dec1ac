"""decode+NMS of the configs[4] share with 16-bit scores: us per call (min of 3 x 20 calls).  With a library built
with ODBOX_TUNING_HOOKS=1 the ODBOX_SCAN_* variables select scan variants (tools/exp_scan16.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import anchors as A, bndbox as BB, workloads   # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    S, B, C = 512, 64, int(os.environ.get('EXP_C', '90'))
    anchors = A.generate_anchors(A.AnchorsConfig(), S)
    N = anchors.shape[0]
    reg_np, sc_np = workloads.nms_inputs(N, C=C, batch=B, seed=1005, p3p4_anchors=9 * (64 * 64 + 32 * 32), distinct=B)
    reg = torch.as_tensor(reg_np).to(dev)
    sc = torch.as_tensor(sc_np).to(dev)
    for name, dt in (('bf16', torch.bfloat16), ('f16', torch.float16), ('f32', torch.float32)):
        x = sc.to(dt)
        for _ in range(3):
            BB.decode_and_nms(anchors, reg, x, (S, S))
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                BB.decode_and_nms(anchors, reg, x, (S, S))
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 50.0)
        print('%s %.1f us' % (name, best), end='   ')
    print(flush=True)


if __name__ == '__main__':
    main()
