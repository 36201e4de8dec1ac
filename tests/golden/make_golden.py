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
from oracle import coco_oracle as CO  # noqa: E402
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


def loss_fixture():
    """metric.py:97-138 on a small batch: losses, normaliser, foreground accuracy hits, gradients."""
    S, B, G, C = 128, 2, 5, 6
    rng = np.random.default_rng(2025)
    anchors = O.generate_anchors(S)
    N = anchors.shape[0]
    gts = [workloads.gt_boxes(rng, S, G, C, pad_to=8) for _ in range(B)]
    gt = np.stack([g[0] for g in gts])
    lab = np.stack([g[1] for g in gts])
    pb = rng.normal(0, 1.5, (B, N, 4)).astype(np.float32)
    ps = rng.uniform(0, 1, (B, N, C)).astype(np.float32)
    ps[0, :40] = 0.0
    ps[0, 40:80] = 1.0
    d, c, n, gb, gs = O.model_metric(anchors, (S, S), gt, lab, pb, ps, C, with_grads=True)
    hits, fg = O.foreground_accuracy(anchors, (S, S), gt, lab, ps, C)
    y = (rng.uniform(size=(B, 7)) < 0.3).astype(np.float32)
    p = rng.uniform(0, 1, (B, 7)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'loss_s128.npz'), image_hw=np.int32([S, S]), num_classes=np.int32(C),
                        anchors=anchors, gt_boxes=gt, gt_labels=lab, pred_boxes=pb, pred_scores=ps,
                        dist_loss=np.float64(d), class_loss=np.float64(c), normalizer=np.float64(n),
                        hits=np.int64(hits), grad_boxes_sum=gb.astype(np.float32), grad_scores_sum=gs.astype(np.float32),
                        image_labels=y, image_scores=p, image_focal_sum=np.float64(O.focal_loss(y, p)))


def coco_fixture():
    """evaluate.py / cocoeval.py on a small gathered batch: the dense match tables and the 12 stats."""
    B, R, G, C = 3, 24, 10, 4
    d = workloads.eval_inputs(B=B, R=R, G=G, C=C, S=256, seed=2026, valid_gt=6, few_classes=3)
    d['dt_scores'][1, :4] = 0.75                     # equal scores: slot order decides
    d['gt_boxes'][2, 1] = d['gt_boxes'][2, 0]        # duplicate GT rows of one class: the later one wins ties
    d['gt_labels'][2, 1] = d['gt_labels'][2, 0]
    r = CO.dense_eval(d['dt_boxes'], d['dt_scores'], d['dt_labels'], d['gt_boxes'], d['gt_labels'], (256, 256), C)
    ev = r['evaluator']
    ev.accumulate()
    ev.summarize()
    np.savez_compressed(os.path.join(HERE, 'coco_eval_small.npz'), image_hw=np.int32([256, 256]), num_classes=np.int32(C),
                        **d, dt_match=r['dt_match'], dt_ignore=r['dt_ignore'], gt_match=r['gt_match'],
                        gt_ignore=r['gt_ignore'], dt_rank=r['dt_rank'], stats=np.asarray(ev.stats, np.float64))


def gt_prepare_fixture():
    """detection.py:97-141,257-262: ragged annotations -> padded batch (fp32 and fp16 pipelines)."""
    rng = np.random.default_rng(2027)
    sizes = [(480, 640), (1080, 1920), (333, 200), (512, 512)]
    boxes, labels = [], []
    for H, W in sizes:
        n = int(rng.integers(0, 9))
        x0, y0 = rng.uniform(0, W - 8, n), rng.uniform(0, H - 8, n)
        boxes.append(np.stack([x0, y0, rng.uniform(1, W - x0), rng.uniform(1, H - y0)], 1).astype(np.float32).reshape(-1, 4))
        labels.append(rng.integers(0, 80, n).astype(np.int32))
    out = {}
    for name, dt in (('f32', np.float32), ('f16', np.float16)):
        ex = [O.prepare_example_boxes(b, l, s, 512, 6, dt) for b, l, s in zip(boxes, labels, sizes)]
        bb, ll = O.padded_batch_boxes(ex, 6)
        out['boxes_' + name], out['labels_' + name] = np.asarray(bb, np.float32), ll
    offs = np.cumsum([0] + [len(b) for b in boxes]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, 'gt_prepare.npz'), image_size=np.int32(512), max_items=np.int32(6),
                        boxes_xywh=np.concatenate(boxes), labels=np.concatenate(labels), row_offsets=offs,
                        orig_hw=np.int32(sizes), **out)


def restore_fixture():
    """infdet.py:140-174: detections back to original image pixels (fp64)."""
    rng = np.random.default_rng(2028)
    B, R = 4, 12
    coords = rng.uniform(0, 512, (B, R, 4)).astype(np.float32)
    scores = np.sort(rng.uniform(0.05, 1, (B, R)).astype(np.float32), axis=1)[:, ::-1].copy()
    for b, n in enumerate((0, 5, 12, 9)):
        scores[b, n:] = 0
    hw = np.int32([(480, 640), (1080, 1920), (333, 200), (512, 512)])
    out, cnt = O.restore_detections(coords, scores, hw, 512)
    np.savez_compressed(os.path.join(HERE, 'restore.npz'), image_size=np.int32(512), coords=coords, scores=scores,
                        orig_hw=hw, boxes=out, counts=cnt)


if __name__ == '__main__':
    match_fixture()
    nms_fixture()
    loss_fixture()
    coco_fixture()
    gt_prepare_fixture()
    restore_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))
