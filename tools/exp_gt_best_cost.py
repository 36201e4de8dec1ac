"""What the per-GT best-anchor output costs: odbox_match_encode with out_gt_best NULL / set, configs[1] and configs[3]."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from object_detection_amd import _lib, anchors as A, workloads
lib = _lib.load(); dev = _lib.require_gpu(); st = _lib.stream_ptr()
for cfg, B, anchors in ((2, 64, bench.yolo_grid_anchors(416)), (4, 128, A.generate_anchors(A.AnchorsConfig(), 512))):
    c = workloads.CONFIGS[cfg]
    gt_np, lab_np = workloads.match_inputs(cfg, batch=B)
    gt, lab = torch.as_tensor(gt_np).to(dev), torch.as_tensor(lab_np).to(dev)
    N, G, C, S = anchors.shape[0], gt.shape[1], c['C'], c['S']
    reg = torch.empty((B, N, 5), device=dev); labels = torch.empty((B, N, C + 1), device=dev)
    best = torch.empty((B, G, 2), dtype=torch.int32, device=dev)
    for name, bp in (('NULL', None), ('set', best.data_ptr()), ('NULL', None), ('set', best.data_ptr())):
        def step(i):
            assert lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(S), float(S), 0.4, 0.5,
                                          reg.data_ptr(), labels.data_ptr(), None, None, None, bp, st) == 0
        _, ev = bench.timed(step, 100, 10, 1)
        print('cfg %d out_gt_best %-4s %.2f us per call' % (cfg, name, ev / 100 * 1e6), flush=True)
