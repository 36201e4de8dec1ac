"""Where does the wall time of bench.py's timed loop go?  host issue time vs HIP-event span vs wall, 1 / 5 output sets."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from object_detection_amd import _lib, workloads
lib = _lib.load(); dev = _lib.require_gpu(); stream = _lib.stream_ptr()
anchors = bench.yolo_grid_anchors(416)
for sets in (1, 5, 1, 5):
    step, w = bench.match_encode_case(lib, dev, stream, 2, 64, anchors, 1002, sets)
    for rep in range(3):
        for i in range(20): step(i)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record()
        for i in range(200): step(20 + i)
        e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print('sets %d rep %d: host issue %.1f us/step, wall %.1f us/step, events %.1f us/step' % (sets, rep, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6, e0.elapsed_time(e1) / 200 * 1e3), flush=True)
    del step, w
