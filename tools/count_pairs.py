#!/usr/bin/env python3
"""(anchor, GT row) pairs the matcher evaluates per launch of a BASELINE configuration: a -DODBOX_COUNT_PAIRS build
(bash tools/build_variant.sh pairs -DODBOX_COUNT_PAIRS; ODBOX_LIBRARY=<that .so>).  The workloads are seeded, so the
counts are constants of (config, tile shape); profiles/pmc_traffic.json keeps the dense-crowd one for bench.py.

    python tools/count_pairs.py [cfg ...]
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _lib, anchors as A, workloads  # noqa: E402


def main():
    lib = _lib.load()
    dev = _lib.require_gpu()
    st = _lib.stream_ptr()
    lib.odbox_set_output_policy(1)                       # the 256-anchor tiles (what configs[3] runs)
    for cfg in [int(x) for x in sys.argv[1:]] or [4]:
        c = workloads.CONFIGS[cfg]
        anchors = A.generate_anchors(A.AnchorsConfig(), c['S'])
        N = anchors.shape[0]
        B = {1: 8, 3: 32, 4: 128}[cfg]
        gt, lab = workloads.match_inputs(cfg, batch=B, seed=1000 + cfg)          # bench.py's seed for rank 0
        gt, lab = torch.as_tensor(gt).to(dev), torch.as_tensor(lab).to(dev)
        G, C = gt.shape[1], c['C']
        reg = torch.empty((B, N, 5), device=dev)
        out = torch.empty((B, N, C + 1), device=dev)
        cnt = (ctypes.c_ulonglong * 4)()
        lib.odbox_debug_pair_count(cnt, 1)
        rc = lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(c['S']), float(c['S']),
                                    0.4, 0.5, reg.data_ptr(), out.data_ptr(), None, None, None, None, st)
        assert rc == 0
        lib.odbox_debug_pair_count(cnt, 1)
        print(json.dumps({'cfg': cfg, 'B': B, 'N': N, 'G': G, 'pairs_evaluated': cnt[0], 'pairs_all': B * N * G,
                          'share': round(cnt[0] / (B * N * G), 4), 'cull_steps': cnt[1], 'waves': cnt[2],
                          'rows_per_wave': round(cnt[0] / 64 / max(cnt[2], 1), 1)}))


if __name__ == '__main__':
    main()
