"""decode+NMS throughput with two batches in flight (configs[4] per-GPU share, 64 images each).

One stream: the 59 us of partition / NMS / finalize run behind the 178 us score scan with the memory
system idle.  Two streams, each with its own workspace and outputs, batches alternating: the tail of
batch k runs beside the scan of batch k+1 (the scan leaves LDS for it, the tail kernels fit in it).
Prints us per 64-image step for 1, 2 and 3 streams and checks that every stream's detections equal the
one-stream result."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _lib, anchors as A, bndbox as BB, workloads   # noqa: E402

if os.environ.get('EXP_OLDLIB'):       # a library built from an older commit (bisecting): it lacks the newest entry point
    _lib.SIGNATURES.pop('odbox_decode_nms_mixed', None)


def main():
    dev = torch.device('cuda', 0)
    if os.environ.get('EXP_FUSED') in ('0', '1'):           # bndbox.set_nms_fused_tail
        BB.set_nms_fused_tail(os.environ['EXP_FUSED'] == '1')
    S, B, C = 512, 64, 90
    anchors = A.generate_anchors(A.AnchorsConfig(), S)
    N = anchors.shape[0]
    reg_np, sc_np = workloads.nms_inputs(N, C=C, batch=B, seed=1005, p3p4_anchors=9 * (64 * 64 + 32 * 32), distinct=B)
    nsets = 3
    regs = [torch.as_tensor(reg_np).to(dev) for _ in range(nsets)]
    scs = [torch.as_tensor(sc_np).to(dev) for _ in range(nsets)]
    if os.environ.get('EXP_DTYPE') in ('bf16', 'f16'):          # 16-bit model outputs (odbox_decode_nms_mixed)
        dt = torch.bfloat16 if os.environ['EXP_DTYPE'] == 'bf16' else torch.float16
        regs, scs = [r.to(dt) for r in regs], [x.to(dt) for x in scs]
    ref = BB.decode_and_nms(anchors, regs[0], scs[0], (S, S))
    torch.cuda.synchronize()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    only = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for nstreams in ((only,) if only else (1, 2, 3)):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        outs = [None] * nstreams

        def run(k):
            for i in range(k):
                j = i % nstreams
                with torch.cuda.stream(streams[j]):
                    outs[j] = BB.decode_and_nms(anchors, regs[j], scs[j], (S, S))
        run(2 * nstreams)
        torch.cuda.synchronize()
        if os.environ.get('EXP_OFFSET') and nstreams > 1:
            # start the streams out of phase (stream j idles j / nstreams of a scan first): do they stay that way?
            for j in range(1, nstreams):
                with torch.cuda.stream(streams[j]):
                    torch.cuda._sleep(int(float(os.environ['EXP_OFFSET']) * j / nstreams))
        best = 1e9
        for _ in range(3):
            if os.environ.get('EXP_OFFSET') and nstreams > 1:
                for j in range(1, nstreams):
                    with torch.cuda.stream(streams[j]):
                        torch.cuda._sleep(int(float(os.environ['EXP_OFFSET']) * j / nstreams))
            t0 = time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps)
        ok = all(all(torch.equal(a, b) for a, b in zip(o, ref)) for o in outs)
        print('%d stream(s): %.1f us per 64-image step, %.0f images/s, detections %s' %
              (nstreams, best * 1e6, B / best, 'identical' if ok else 'DIFFER'), flush=True)


if __name__ == '__main__':
    main()
