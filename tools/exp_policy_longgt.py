"""Through or around the caches for GT lists longer than the sweep takes (G = 100, 200; C = 80): the streaming tiles against the
256-anchor tiles, one buffer in place and five in rotation."""
import os, sys
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from object_detection_amd import _lib, workloads
lib = _lib.load(); dev = _lib.require_gpu(); st = _lib.stream_ptr()
anchors = bench.yolo_grid_anchors(416)
N, S, C = anchors.shape[0], 416, 80
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for G in [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (65, 100, 200):
    rng = np.random.default_rng(G)
    gts = [workloads.gt_boxes(rng, S, G, C) for _ in range(B)]
    gt = torch.as_tensor(np.stack([g[0] for g in gts])).to(dev)
    lab = torch.as_tensor(np.stack([g[1] for g in gts])).to(dev)
    for sets in ((1, 5) if B <= 64 else (1,)):
        regs = [torch.empty((B, N, 5), device=dev) for _ in range(sets)]
        labs = [torch.empty((B, N, C + 1), device=dev) for _ in range(sets)]
        def step(i):
            k = i % sets
            assert lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(S), float(S), 0.4, 0.5,
                                          regs[k].data_ptr(), labs[k].data_ptr(), None, None, None, None, st) == 0
        out = []
        for pol in (1, 2, 0, 1, 2, 0):
            lib.odbox_set_output_policy(pol)
            it = 60 if B <= 64 else 20
            _, ev = bench.timed(step, it, 5, 1)
            out.append(ev / it * 1e6)
        lib.odbox_set_output_policy(0)
        print('G %3d x %d set(s): cache %.1f / %.1f   stream %.1f / %.1f   auto %.1f / %.1f us (plan %d)'
              % (G, sets, out[0], out[3], out[1], out[4], out[2], out[5], lib.odbox_match_encode_last_plan()), flush=True)
        del regs, labs
        torch.cuda.empty_cache()
