"""One mode of bench.py's headline workload (configs[1] match+encode), for the PMC passes of tools/profile_round.sh:
    python3 tools/pmc_headline.py stream|auto|resident [steps]
stream / auto: five output sets in rotation under that output policy; resident: one set rewritten in place under 'cache'."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from object_detection_amd import _lib, workloads  # noqa: E402

mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lib = _lib.load()
dev = _lib.require_gpu()
anchors = bench.yolo_grid_anchors(416)
step, w = bench.match_encode_case(lib, dev, _lib.stream_ptr(), 2, workloads.CONFIGS[2]['B'], anchors, 1002,
                                  1 if mode == 'resident' else bench.OUTPUT_SETS)
lib.odbox_set_output_policy({'stream': 2, 'auto': 0, 'resident': 1}[mode])      # resident: through the caches ('cache')
for i in range(steps):
    step(i)
torch.cuda.synchronize()
print(mode, 'done')
