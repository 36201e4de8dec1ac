"""Host time to ISSUE one decode+NMS batch (nothing waits for the GPU): plain call in a stream context against
DecodeNmsPipeline.submit + result.  If it is above the GPU's time per batch the pipeline is host-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import anchors as A, bndbox as BB, workloads   # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    S, B, C = 512, 64, 90
    anchors = A.generate_anchors(A.AnchorsConfig(), S)
    N = anchors.shape[0]
    reg_np, sc_np = workloads.nms_inputs(N, C=C, batch=B, seed=1005, p3p4_anchors=9 * (64 * 64 + 32 * 32), distinct=B)
    reg = torch.as_tensor(reg_np).to(dev).to(torch.bfloat16)
    sc = torch.as_tensor(sc_np).to(dev).to(torch.bfloat16)
    streams = [torch.cuda.Stream() for _ in range(3)]
    pipe = BB.DecodeNmsPipeline(depth=3)
    for _ in range(6):
        pipe.submit(anchors, reg, sc, (S, S), inputs_ready=True).result()
    torch.cuda.synchronize()
    K = 30

    def plain():
        for i in range(K):
            with torch.cuda.stream(streams[i % 3]):
                BB.decode_and_nms(anchors, reg, sc, (S, S))

    def piped():
        pend = []
        for i in range(K):
            pend.append(pipe.submit(anchors, reg, sc, (S, S), inputs_ready=True))
            if len(pend) == 3:
                pend.pop(0).result()
        for p in pend:
            p.result()

    for name, fn in (('plain call in a stream context', plain), ('pipeline submit + result', piped)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('%-34s issue %.1f us per batch, until done %.1f us per batch' % (name, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))


if __name__ == '__main__':
    main()
