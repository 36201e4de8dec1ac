import os, sys
import torch, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from object_detection_amd import _lib, workloads
lib = _lib.load(); dev = _lib.require_gpu(); st = _lib.stream_ptr()
anchors = bench.yolo_grid_anchors(416)
N, S = anchors.shape[0], 416
for C, B, sets in ((20, 256, 1), (20, 256, 5), (20, 64, 1), (20, 64, 8), (40, 128, 1), (40, 128, 5), (80, 64, 5)):
    gt_np, lab_np = workloads.match_inputs(2, batch=B)
    lab_np = (lab_np % C).astype(np.int32)
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
    print('C %d B %3d (%6.1f MB of labels) x %d set(s): cache %.1f / %.1f   stream %.1f / %.1f   auto %.1f / %.1f us (plan %d)'
          % (C, B, mb, sets, out[0], out[3], out[1], out[4], out[2], out[5], lib.odbox_match_encode_last_plan()), flush=True)
    del regs, labs
    torch.cuda.empty_cache()
