"""Regenerates the golden fixtures in this directory from the NumPy oracle
(oracle/odbox_oracle.py).  The reference itself cannot be executed here (TensorFlow absent,
SURVEY.md §8c) and holds no fixtures for this path, so these vectors pin the build's own
oracle ("parity unpinned" against the reference); the C oracle and the HIP path are both
checked against them.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from object_detection_amd import workloads  # noqa: E402
from oracle import odbox_oracle as O  # noqa: E402


def match_fixture():
    S, B, G, C = 128, 3, 5, 6
    rng = np.random.default_rng(2024)
    anchors = O.generate_anchors(S)
    gts = [workloads.gt_boxes(rng, S, G, C, pad_to=8) for _ in range(B)]
    gt = np.stack([g[0] for g in gts])
    lab = np.stack([g[1] for g in gts])
    gt[0, 1] = anchors[100]          # an exact match (IoU 1)
    gt[0, 2] = anchors[100]          # and its duplicate (argmax tie -> lower row)
    gt[2] = -1                       # an image with only padding
    lab[2] = -1
    r = O.anchor_targets_bbox(anchors, (S, S - 16), gt, lab, C)
    np.savez_compressed(os.path.join(HERE, 'match_encode_s128.npz'), image_hw=np.int32([S, S - 16]),
                        num_classes=np.int32(C), anchors=anchors, gt_boxes=gt, gt_labels=lab, **r)


def nms_fixture():
    S, B, C = 128, 2, 6
    anchors = O.generate_anchors(S)
    N = anchors.shape[0]
    reg, sc = workloads.nms_inputs(N, C=C, batch=B, seed=77, clusters=12, distinct=B,
                                   p3p4_anchors=9 * (16 * 16 + 8 * 8))
    sc[1, 5:9, 3] = 0.75             # equal scores on neighbouring anchors
    sc[1, 5, 1] = 0.75               # and an argmax tie inside one row
    boxes, scores, labels, keep, count = O.decode_and_nms(anchors, reg, sc, (S, S), max_ret=40)
    np.savez_compressed(os.path.join(HERE, 'decode_nms_s128.npz'), image_hw=np.int32([S, S]),
                        max_ret=np.int32(40), anchors=anchors, regressors=reg, class_scores=sc,
                        boxes=boxes, scores=scores, labels=labels, keep_anchor=keep, count=count)


if __name__ == '__main__':
    match_fixture()
    nms_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))
