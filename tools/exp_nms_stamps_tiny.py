import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
from object_detection_amd import _lib, anchors as A, bndbox as BB
import exp_nms_overlap as E, exp_nms_stamps as ST
lib = _lib.load(); dev = _lib.require_gpu()
S, B, C = 512, 64, 90
anchors = A.generate_anchors(A.AnchorsConfig(), S); an = anchors.cpu().numpy()
rng = np.random.default_rng(7)
reg8, sc8 = E.make(an, 8, C, int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 0.03, rng)
reg = torch.as_tensor(np.tile(reg8, (8, 1, 1))).to(dev); sc = torch.as_tensor(np.tile(sc8, (8, 1, 1))).to(dev)
arr = (ctypes.c_ulonglong * 64); s, m, n = arr(), arr(), arr()
for _ in range(3): BB.decode_and_nms(anchors, reg, sc, (S, S))
lib.odbox_debug_stamps(s, m, n, 1)
BB.decode_and_nms(anchors, reg, sc, (S, S))
lib.odbox_debug_stamps(s, m, n, 1)
ST.NAMES.update(ST.FUSED)
for i in range(64):
    if n[i]: print('%-42s %6d  %7.2f  %7.2f' % (ST.NAMES.get(i, 'slot %d' % i), n[i], s[i] / n[i] / 100.0, m[i] / 100.0))
