"""Group-of-10 timings of the headline loop (what bench.py's per_step_median sees), 60 groups."""
import os, sys, gc
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from object_detection_amd import _lib
lib = _lib.load(); dev = _lib.require_gpu(); st = _lib.stream_ptr()
anchors = bench.yolo_grid_anchors(416)
step, w = bench.match_encode_case(lib, dev, st, 2, 64, anchors, 1002, bench.OUTPUT_SETS)
lib.odbox_set_output_policy(2)
for rep in range(3):
    groups, per = 60, 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    gc.disable()
    for g in range(groups):
        ev[g].record()
        for i in range(per):
            step(g * per + i)
    ev[groups].record()
    torch.cuda.synchronize()
    gc.enable()
    ts = [ev[g].elapsed_time(ev[g + 1]) * 1e3 / per for g in range(groups)]
    print(' '.join('%.1f' % t for t in ts), flush=True)
