"""How fast can this GPU write a buffer of the bench's output size at all?  Times torch's fill
kernel and hipMemsetAsync (through torch) on 234 MB / 245 MB and on 2.2 GB, for comparison with
match_encode_kernel's store stream."""
import json

import torch

dev = torch.device('cuda')


def timed(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for mb in (234, 245, 490, 2262):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    t_fill = timed(lambda: x.fill_(1.0))
    t_zero = timed(lambda: x.zero_())
    y = torch.empty(n // 2, dtype=torch.float32, device=dev)
    z = torch.empty(n // 2, dtype=torch.float32, device=dev)
    t_copy = timed(lambda: z.copy_(y))
    print(json.dumps(dict(MB=mb, fill_us=t_fill * 1e6, fill_TBs=n * 4 / t_fill / 1e12, zero_us=t_zero * 1e6,
                          zero_TBs=n * 4 / t_zero / 1e12, copy_us=t_copy * 1e6, copy_TBs=n * 4 / t_copy / 1e12)))
    del x, y, z

# read ceiling for the loss kernels: a plain sum over the bytes the forward pass reads
for mb in (120, 240, 480):
    n = mb * 1000 * 1000 // 4
    x = torch.rand(n, dtype=torch.float32, device=dev)
    t = timed(lambda: x.sum())
    t2 = timed(lambda: x.max())
    print(json.dumps(dict(MB=mb, sum_us=t * 1e6, sum_TBs=n * 4 / t / 1e12, max_us=t2 * 1e6, max_TBs=n * 4 / t2 / 1e12)))
    del x
