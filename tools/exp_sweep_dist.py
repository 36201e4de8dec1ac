"""Per-launch time distribution of the configs[1] headline (rotating sets, policy stream); tuning-hooks build:
ODBOX_SWEEP_XCD_RUNS=0/1 switches the piece assignment."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from object_detection_amd import _lib, workloads
lib = _lib.load(); dev = _lib.require_gpu(); st = _lib.stream_ptr()
anchors = bench.yolo_grid_anchors(416)
step, w = bench.match_encode_case(lib, dev, st, 2, 64, anchors, 1002, bench.OUTPUT_SETS)
lib.odbox_set_output_policy(2)
n = 400
for rep in range(2):
    for xr in ('1', '0'):
        os.environ['ODBOX_SWEEP_XCD_RUNS'] = xr
        for i in range(50):
            step(i)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        torch.cuda.synchronize()
        for i in range(n):
            ev[i].record()
            step(i)
        ev[n].record()
        torch.cuda.synchronize()
        t = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)])
        total = ev[0].elapsed_time(ev[n]) * 1e3 / n
        print('xcd_runs %s: mean %.2f (whole loop %.2f)  min %.2f  p10 %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f   by set: %s'
              % (xr, t.mean(), total, t.min(), np.percentile(t, 10), np.median(t), np.percentile(t, 90), np.percentile(t, 99), t.max(),
                 ' '.join('%.1f' % t[k::bench.OUTPUT_SETS].mean() for k in range(bench.OUTPUT_SETS))), flush=True)
