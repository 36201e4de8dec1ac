"""Synthetic workloads of the five BASELINE.json configurations (recipes: SURVEY.md §8d).
NumPy `default_rng(1000 + cfg)`; anchors come from the caller (HIP generator in bench.py, the
oracle's generator in the CPU tests) so this module computes nothing on the path itself."""
import math

import numpy as np

CONFIGS = {
    1: dict(name='cfg1_effdet_d0_512', S=512, B=8, G=20, C=80, grid='retina'),
    2: dict(name='cfg2_yolov3_416', S=416, B=64, G=20, C=80, grid='yolo'),
    3: dict(name='cfg3_effdet_d4_1024', S=1024, B=256, G=20, C=80, grid='retina'),
    4: dict(name='cfg4_dense_crowd', S=512, B=128, G=500, C=1, grid='retina'),
    5: dict(name='cfg5_decode_nms', S=512, B=512, G=0, C=90, grid='retina'),
}


def num_anchors(cfg):
    c = CONFIGS[cfg]
    if c['grid'] == 'yolo':
        return 3 * sum((c['S'] // s) ** 2 for s in (8, 16, 32))
    return 9 * sum((c['S'] // 2 ** l) ** 2 for l in range(3, 8))


def match_bytes_per_image(N, G, C):
    """ALGORITHMIC bytes of match+encode per image:
    anchors read 16N + GT read 20G + regression written 20N + labels written 4(C+1)N."""
    return N * (36 + 4 * (C + 1)) + 20 * G


def nms_bytes_per_image(N, C, max_ret=100):
    """ALGORITHMIC bytes of decode+NMS per image: anchors 16N + regressors 16N + scores 4CN +
    outputs 24*max_ret."""
    return N * (32 + 4 * C) + 24 * max_ret


def gt_boxes(rng, S, G, C, pad_to=None):
    """One image of cfg 1-3 ground truth: G valid rows (+ -1 padding up to pad_to)."""
    boxes = np.empty((G, 4), dtype=np.float32)
    for g in range(G):
        while True:
            cx, cy = rng.uniform(0, S, size=2)
            w, h = np.exp(rng.uniform(math.log(8), math.log(S / 2), size=2))
            x0, y0 = max(cx - w / 2, 0.0), max(cy - h / 2, 0.0)
            x1, y1 = min(cx + w / 2, S - 1.0), min(cy + h / 2, S - 1.0)
            if x1 - x0 >= 1 and y1 - y0 >= 1:
                break
        boxes[g] = (x0, y0, x1, y1)
    labels = rng.integers(0, C, size=G).astype(np.int32)
    if pad_to is not None and pad_to > G:
        boxes = np.concatenate([boxes, np.full((pad_to - G, 4), -1, np.float32)])
        labels = np.concatenate([labels, np.full((pad_to - G,), -1, np.int32)])
    return boxes, labels


def crowd_boxes(rng, S, G):
    """One image of cfg 4: 10 cluster centres, boxes scattered N(0, 40^2) around them."""
    centres = rng.uniform(0, S, size=(10, 2))
    which = rng.integers(0, 10, size=G)
    c = centres[which] + rng.normal(0, 40, size=(G, 2))
    wh = np.exp(rng.uniform(math.log(8), math.log(64), size=(G, 2)))
    b = np.stack([c[:, 0] - wh[:, 0] / 2, c[:, 1] - wh[:, 1] / 2,
                  c[:, 0] + wh[:, 0] / 2, c[:, 1] + wh[:, 1] / 2], axis=1)
    b = np.clip(b, 0, S - 1.0).astype(np.float32)
    bad = (b[:, 2] - b[:, 0] < 1) | (b[:, 3] - b[:, 1] < 1)
    b[bad] = np.array([0, 0, 8, 8], np.float32)
    return b, np.zeros((G,), np.int32)


def match_inputs(cfg, batch=None, pad_to=None, seed=None):
    """(gt_boxes [B,G',4] f32, gt_labels [B,G'] i32) for cfg 1-4."""
    c = CONFIGS[cfg]
    B = c['B'] if batch is None else batch
    rng = np.random.default_rng(1000 + cfg if seed is None else seed)
    bs, ls = [], []
    for _ in range(B):
        if cfg == 4:
            b, l = crowd_boxes(rng, c['S'], c['G'])
        else:
            b, l = gt_boxes(rng, c['S'], c['G'], c['C'], pad_to)
        bs.append(b)
        ls.append(l)
    return np.stack(bs), np.stack(ls)


def nms_inputs(N, C=90, batch=64, seed=1005, clusters=50, p3p4_anchors=None, distinct=8):
    """cfg 5 inputs: regressors [B,N,4], class_scores [B,N,C].  Background scores U(0,0.25);
    `clusters` runs of 90 consecutive anchors (start a multiple of 9 inside P3/P4) whose members
    score 0.3 + 0.7 U quantised to 2^-10 on the cluster class (p = 0.9) or a random class.
    To bound host time only `distinct` different images are drawn and then repeated."""
    rng = np.random.default_rng(seed)
    if p3p4_anchors is None:
        p3p4_anchors = N
    d = min(distinct, batch)
    reg = np.clip(rng.normal(0, 0.5, size=(d, N, 4)), -2, 2).astype(np.float32)
    sc = rng.uniform(0, 0.25, size=(d, N, C)).astype(np.float32)
    for b in range(d):
        for _ in range(clusters):
            start = int(rng.integers(0, max(1, (p3p4_anchors - 90) // 9))) * 9
            cls = int(rng.integers(0, C))
            for a in range(start, min(start + 90, N)):
                k = cls if rng.uniform() < 0.9 else int(rng.integers(0, C))
                s = 0.3 + 0.7 * rng.uniform()
                sc[b, a, k] = np.float32(math.floor(s * 1024) / 1024)
    reps = (batch + d - 1) // d
    reg = np.tile(reg, (reps, 1, 1))[:batch]
    sc = np.tile(sc, (reps, 1, 1))[:batch]
    return reg, sc


def eval_inputs(B=64, R=100, G=64, C=80, S=512, seed=1006, valid_gt=20, few_classes=None):
    """Evaluation-side batch: padded GT `[B,G,4]` / `[B,G]` (valid_gt real rows,
    the rest -1) and padded, score-sorted detections `[B,R,4]`, `[B,R]`, `[B,R]` shaped like the
    output of decode+NMS: jittered copies of GT rows (some twice, so duplicates compete for one
    GT), false positives, exact score ties (scores on a 2^-7 grid), then zero rows.
    few_classes: draw labels from that many classes only (forces several GT per category)."""
    rng = np.random.default_rng(seed)
    cls = C if few_classes is None else few_classes
    gt_b = np.full((B, G, 4), -1, np.float32)
    gt_l = np.full((B, G), -1, np.int32)
    dt_b = np.zeros((B, R, 4), np.float32)
    dt_s = np.zeros((B, R), np.float32)
    dt_l = np.zeros((B, R), np.int32)
    for b in range(B):
        nv = min(valid_gt, G)
        boxes, _ = gt_boxes(rng, S, nv, C)
        labels = rng.integers(0, cls, size=nv).astype(np.int32)
        gt_b[b, :nv] = boxes
        gt_l[b, :nv] = labels
        n_det = int(rng.integers(R // 2, R + 1)) if nv else 0
        rows = []
        for _ in range(n_det):
            if nv and rng.random() < 0.7:
                g = int(rng.integers(0, nv))
                jit = rng.normal(0, 0.08, size=4) * np.tile(boxes[g, 2:] - boxes[g, :2], 2)
                box = boxes[g] + jit.astype(np.float32)
                lab = labels[g] if rng.random() < 0.9 else rng.integers(0, cls)
            else:
                box, lab = gt_boxes(rng, S, 1, C)[0][0], rng.integers(0, cls)
            box = np.array([min(box[0], box[2]), min(box[1], box[3]), max(box[0], box[2]), max(box[1], box[3])])
            rows.append((np.round(rng.uniform(0.3, 1.0) * 128) / 128, box, lab))
        rows.sort(key=lambda r: -r[0])
        for k, (s, box, lab) in enumerate(rows):
            dt_s[b, k], dt_b[b, k], dt_l[b, k] = s, box, lab
    return dict(dt_boxes=dt_b, dt_scores=dt_s, dt_labels=dt_l, gt_boxes=gt_b, gt_labels=gt_l)
