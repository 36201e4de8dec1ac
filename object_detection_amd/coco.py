"""Drop-in for the box arithmetic of the reference's legacy NumPy pipeline (coco.py) on MI355X
(SURVEY.md §8f row 4): `calc_iou` (coco.py:169-191) and the sequential per-GT anchor matcher inside
`COCO_Iterable.process_image` (coco.py:379-441).  HIP only, no CPU fallback.

The reference runs these per image on NumPy float64 values; here a whole batch goes through one
launch in float64 with the same association, so results are bit-identical to NumPy's.  The dtype of
`np_anchor_boxes` is left open by the reference (only `complete_initialization`, which nothing
calls, sets it): float64 is taken.  COCO json reading, cv2 and albumentations stay out of scope.
"""
import ctypes

import torch

from . import _lib


class ProcessingError(Exception):
    """coco.py:18-20."""

    def __init__(self, message):
        super().__init__(message)
        self.message = message


def calc_iou(box, boxes, area):
    """coco.py:169-191: box (cx, cy, h, w) against boxes [N,4] in the same layout with their areas
    [N]; +1 pixel on the intersection only.  box may also be [K,4]: K rows of results."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    a = _lib.as_dev(box, torch.float64, dev)
    single = a.dim() == 1
    a = a.reshape(-1, 4)
    b = _lib.as_dev(boxes, torch.float64, dev).reshape(-1, 4)
    ar = _lib.as_dev(area, torch.float64, dev).reshape(-1)
    if ar.numel() != b.shape[0]:
        raise ValueError('calc_iou: %d areas for %d boxes' % (ar.numel(), b.shape[0]))
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=dev)
    _lib.check(lib.odbox_legacy_iou(_lib.ptr(a), _lib.ptr(b), _lib.ptr(ar), a.shape[0], b.shape[0], 0, _lib.ptr(out),
                                    _lib.stream_ptr()), 'odbox_legacy_iou')
    return out[0] if single else out


def match_anchors(np_anchor_boxes, np_anchor_areas, bboxes, labels, max_iou_threshold=0.5, return_max_ious=False):
    """coco.py:379-441 for a batch.  np_anchor_boxes [N,4] (cx, cy, h, w), np_anchor_areas [N];
    bboxes [B,G,4] rows of (x0, y0, w, h) in visiting order (rows with w <= 3 or h <= 3, and the -1
    padding rows, are skipped like :418-419); labels [B,G] the converted categories.
    Returns (true_bboxes [B,N,4] float64, true_labels [B,N] float64, label_sum [B]); an image whose
    label_sum is 0 is the reference's ProcessingError('there are no true labels') (:439-462) --
    `check_true_labels(label_sum)` raises it."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    an = _lib.as_dev(np_anchor_boxes, torch.float64, dev).reshape(-1, 4)
    ar = _lib.as_dev(np_anchor_areas, torch.float64, dev).reshape(-1)
    gt = _lib.as_dev(bboxes, torch.float64, dev)
    lab = _lib.as_dev(labels, torch.int32, dev)
    if gt.dim() == 2:
        gt, lab = gt[None], lab[None]
    B, G = gt.shape[0], gt.shape[1]
    N = an.shape[0]
    if ar.numel() != N or lab.shape != (B, G) or gt.shape[2] != 4:
        raise ValueError('match_anchors: shapes %s %s %s %s' % (tuple(an.shape), tuple(ar.shape), tuple(gt.shape), tuple(lab.shape)))
    out_b = torch.empty((B, N, 4), dtype=torch.float64, device=dev)
    out_l = torch.empty((B, N), dtype=torch.float64, device=dev)
    out_m = torch.empty((B, N), dtype=torch.float32, device=dev) if return_max_ious else None
    out_s = torch.empty((B,), dtype=torch.float64, device=dev)
    _lib.check(lib.odbox_legacy_match(_lib.ptr(an), _lib.ptr(ar), _lib.ptr(gt), _lib.ptr(lab), B, N, G,
                                      ctypes.c_double(max_iou_threshold), _lib.ptr(out_b), _lib.ptr(out_l), _lib.ptr(out_m),
                                      _lib.ptr(out_s), _lib.stream_ptr()), 'odbox_legacy_match')
    if return_max_ious:
        return out_b, out_l, out_s, out_m
    return out_b, out_l, out_s


def check_true_labels(label_sum):
    """coco.py:439-462: an image none of whose anchors got a label is an error (one host sync)."""
    bad = (label_sum == 0).nonzero().flatten().tolist()
    if bad:
        raise ProcessingError('there are no true labels')
