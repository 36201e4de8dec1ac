"""Which output policy wins where: configs[1] grid at several batch sizes, one buffer rewritten in place or several in
rotation, policy 'cache' (256-anchor tiles through the caches) against 'stream' (sweep / streaming tiles).
    python tools/exp_policy.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from object_detection_amd import _lib, anchors as A, workloads
lib = _lib.load(); dev = _lib.require_gpu(); st = _lib.stream_ptr()
anchors = bench.yolo_grid_anchors(416)
N, C, S = anchors.shape[0], 80, 416
cases = [(8, 1), (8, 4), (8, 12), (16, 1), (16, 4), (16, 8), (32, 1), (32, 2), (32, 4), (64, 1), (64, 2), (64, 5), (100, 1), (100, 2), (140, 1), (140, 2)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(':')) for a in sys.argv[1:]]
for B, sets in cases:
    gt_np, lab_np = workloads.match_inputs(2, batch=B)
    gt, lab = torch.as_tensor(gt_np).to(dev), torch.as_tensor(lab_np).to(dev)
    G = gt.shape[1]
    regs = [torch.empty((B, N, 5), device=dev) for _ in range(sets)]
    labs = [torch.empty((B, N, C + 1), device=dev) for _ in range(sets)]
    def step(i):
        k = i % sets
        assert lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(S), float(S), 0.4, 0.5,
                                      regs[k].data_ptr(), labs[k].data_ptr(), None, None, None, None, st) == 0
    out = []
    for pol in (1, 2, 0, 1, 2, 0):
        lib.odbox_set_output_policy(pol)
        _, ev = bench.timed(step, 60, 10, 1)
        out.append(ev / 60 * 1e6)
    lib.odbox_set_output_policy(0)
    mb = B * N * (C + 1) * 4 / 1e6
    print('B %3d (%6.1f MB of labels) x %2d set(s) = %6.1f MB in rotation: cache %.1f / %.1f   stream %.1f / %.1f   auto %.1f / %.1f us'
          % (B, mb, sets, mb * sets, out[0], out[3], out[1], out[4], out[2], out[5]), flush=True)
    del regs, labs
    torch.cuda.empty_cache()
