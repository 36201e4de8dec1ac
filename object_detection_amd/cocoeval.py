"""Drop-in for the evaluation tail of the reference (evaluate.py:14-55,93-110,172-190 and
cocoeval.py) on MI355X, fed with tensors instead of COCO dict lists.

Device work (HIP only, no CPU fallback): `coco_records` (xyxy -> xywh / areas), `compute_iou`
(cocoeval.py:167-194, fp64 like pycocotools' bbIou) and `evaluate_img` (cocoeval.py:239-317 for
every image, category, area range and IoU threshold in one launch).  `COCOeval.accumulate` /
`.summarize` (cocoeval.py:319-497) are the reference's own host-side NumPy bookkeeping over the
match tables and stay on the host here as well.

    ev = COCOeval(num_classes)
    for batch: ev.add_batch(boxes, scores, labels, true_bboxes, true_labels)   # evaluate.py:83-110
    ev.accumulate(); ev.summarize(); ev.stats[0]                               # evaluate.py:186-190
"""
import ctypes

import numpy as np
import torch

from . import _lib


class Params:
    """cocoeval.py:506-515 (bbox parameters)."""

    def __init__(self):
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ['all', 'small', 'medium', 'large']
        self.useCats = 1


def coco_records(boxes, want_area_f64=False):
    """evaluate.py:19-22,39-45: boxes [...,4] xyxy -> (xywh [...,4], area [...]).  The area is the
    fp32 `b_h * b_w` of a GT annotation (:41), or with want_area_f64 the detection's double
    `bb[2]*bb[3]` (:177)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    boxes = _lib.as_dev(boxes, torch.float32, dev)
    if boxes.shape[-1] != 4:
        raise ValueError('boxes must be [...,4], got %r' % (tuple(boxes.shape),))
    n = boxes.numel() // 4
    xywh = torch.empty_like(boxes)
    area = torch.empty(boxes.shape[:-1], dtype=torch.float64 if want_area_f64 else torch.float32, device=dev)
    _lib.check(lib.odbox_coco_records(_lib.ptr(boxes), n, _lib.ptr(xywh),
                                      None if want_area_f64 else _lib.ptr(area),
                                      _lib.ptr(area) if want_area_f64 else None, _lib.stream_ptr()),
               'odbox_coco_records')
    return xywh, area


def compute_iou(dt_boxes, gt_boxes):
    """cocoeval.py:167-194 for whole batches: dt_boxes [B,R,4], gt_boxes [B,G,4] xyxy ->
    [B,R,G] float64 (maskUtils.iou semantics, iscrowd = 0).  `COCOeval.ious[img, cat]` of the
    reference is the sub-matrix of that category's detection slots (score order) and GT rows."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    dt_boxes = _lib.as_dev(dt_boxes, torch.float32, dev)
    gt_boxes = _lib.as_dev(gt_boxes, torch.float32, dev)
    if dt_boxes.dim() != 3 or gt_boxes.dim() != 3 or dt_boxes.shape[0] != gt_boxes.shape[0]:
        raise ValueError('need dt_boxes [B,R,4] and gt_boxes [B,G,4]')
    B, R, _ = dt_boxes.shape
    G = gt_boxes.shape[1]
    out = torch.empty((B, R, G), dtype=torch.float64, device=dev)
    _lib.check(lib.odbox_coco_iou(_lib.ptr(dt_boxes), _lib.ptr(gt_boxes), B, R, G, _lib.ptr(out),
                                  _lib.stream_ptr()), 'odbox_coco_iou')
    return out


def evaluate_img(dt_boxes, dt_scores, dt_labels, gt_boxes, gt_labels, num_classes, params=None):
    """cocoeval.py:239-317 for every (image, category, area range, IoU threshold) of a batch.

    dt_* are the padded detections `[B,R,4]`, `[B,R]`, `[B,R]` (padding rows are evaluated as the
    category-0, score-0 detections they are, evaluate.py:104-110); gt_* the padded ground truth
    `[B,G,4]`, `[B,G]` (label -1 rows dropped, evaluate.py:95-98).  Returns a dict of device
    tensors: dt_match [B,A,T,R] i32 (1 + GT row, 0 none), dt_ignore [B,A,T,R] u8, gt_match
    [B,A,T,G] i32 (1 + slot), gt_ignore [B,A,G] u8 (255 = absent row), dt_rank [B,R] i32
    (position inside the (image, category) list, -1 = not evaluated)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    p = params or Params()
    dt_boxes = _lib.as_dev(dt_boxes, torch.float32, dev)
    dt_scores = _lib.as_dev(dt_scores, torch.float32, dev)
    dt_labels = _lib.as_dev(dt_labels, torch.int32, dev)
    gt_boxes = _lib.as_dev(gt_boxes, torch.float32, dev)
    gt_labels = _lib.as_dev(gt_labels, torch.int32, dev)
    if dt_boxes.dim() != 3 or dt_boxes.shape[-1] != 4 or tuple(dt_scores.shape) != tuple(dt_boxes.shape[:2]) \
            or tuple(dt_labels.shape) != tuple(dt_boxes.shape[:2]):
        raise ValueError('need dt_boxes [B,R,4], dt_scores [B,R], dt_labels [B,R]')
    B, R, _ = dt_boxes.shape
    if gt_boxes.dim() != 3 or gt_boxes.shape[0] != B or tuple(gt_labels.shape) != tuple(gt_boxes.shape[:2]):
        raise ValueError('need gt_boxes [B,G,4], gt_labels [B,G]')
    G = gt_boxes.shape[1]
    thr = np.ascontiguousarray(p.iouThrs, dtype=np.float64)
    rng = np.ascontiguousarray(p.areaRng, dtype=np.float64).reshape(-1, 2)
    T, A = len(thr), len(rng)
    out = dict(dt_match=torch.empty((B, A, T, R), dtype=torch.int32, device=dev),
               dt_ignore=torch.empty((B, A, T, R), dtype=torch.uint8, device=dev),
               gt_match=torch.empty((B, A, T, G), dtype=torch.int32, device=dev),
               gt_ignore=torch.empty((B, A, G), dtype=torch.uint8, device=dev),
               dt_rank=torch.empty((B, R), dtype=torch.int32, device=dev))
    need = lib.odbox_coco_match_workspace(B, R)
    ws = _lib.workspace('coco_match', dev, need)
    _lib.check(lib.odbox_coco_match(
        _lib.ptr(dt_boxes), _lib.ptr(dt_scores), _lib.ptr(dt_labels), _lib.ptr(gt_boxes), _lib.ptr(gt_labels),
        B, R, G, int(num_classes), thr.ctypes.data_as(ctypes.c_void_p), T, rng.ctypes.data_as(ctypes.c_void_p), A,
        int(sorted(p.maxDets)[-1]), _lib.ptr(out['dt_match']), _lib.ptr(out['dt_ignore']), _lib.ptr(out['gt_match']),
        _lib.ptr(out['gt_ignore']), _lib.ptr(out['dt_rank']), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
        'odbox_coco_match')
    return out


class COCOeval:
    """COCOeval(cocoGt, cocoDt, 'bbox') of the reference, fed batch by batch with the tensors the
    detector and the input pipeline already hold.  evaluate() of the reference = add_batch() here
    (the per-image matching runs on the GPU as the batches arrive); accumulate() / summarize() /
    stats keep their names and meaning."""

    def __init__(self, num_classes, params=None):
        self.num_classes = int(num_classes)
        self.params = params or Params()
        self.params.maxDets = sorted(self.params.maxDets)                  # cocoeval.py:141
        self._batches = []
        self.eval = {}
        self.stats = []

    def add_batch(self, bboxes, scores, categories, true_bboxes, true_labels):
        """evaluate.py:85-110 for one batch: `bboxes, scores, categories` as the model returns them."""
        r = evaluate_img(bboxes, scores, categories, true_bboxes, true_labels, self.num_classes, self.params)
        dev = r['dt_rank'].device
        self._batches.append(dict(
            dt_match=r['dt_match'], dt_ignore=r['dt_ignore'], gt_ignore=r['gt_ignore'], dt_rank=r['dt_rank'],
            dt_scores=_lib.as_dev(scores, torch.float32, dev), dt_labels=_lib.as_dev(categories, torch.int32, dev),
            gt_labels=_lib.as_dev(true_labels, torch.int32, dev)))
        return r

    def _host(self, name):
        return np.concatenate([b[name].cpu().numpy() for b in self._batches], axis=0)

    def accumulate(self):
        """cocoeval.py:319-424 on the dense match tables."""
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), self.num_classes, len(p.areaRng), len(p.maxDets)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        scores = -np.ones((T, R, K, A, M))
        if self._batches:
            dt_match, dt_ignore = self._host('dt_match'), self._host('dt_ignore')
            gt_ignore, dt_rank = self._host('gt_ignore'), self._host('dt_rank')
            dt_scores, dt_labels, gt_labels = self._host('dt_scores'), self._host('dt_labels'), self._host('gt_labels')
            img, slot = np.nonzero(dt_rank >= 0)
            det_cat = dt_labels[img, slot]
            det_rank = dt_rank[img, slot]
            # per category: detections in (image, rank) order = the concatenation order of :366
            by_cat = np.lexsort((det_rank, img, det_cat))
            img, slot, det_cat, det_rank = img[by_cat], slot[by_cat], det_cat[by_cat], det_rank[by_cat]
            cat_start = np.searchsorted(det_cat, np.arange(K + 1))
            for k in range(K):
                sl = slice(cat_start[k], cat_start[k + 1])
                ki, ks, kr = img[sl], slot[sl], det_rank[sl]
                gt_of_k = gt_labels == k                                     # [I,G]
                for a in range(A):
                    gig = gt_ignore[:, a, :][gt_of_k]
                    npig = np.count_nonzero(gig == 0)
                    for m, max_det in enumerate(p.maxDets):
                        if len(ki) == 0 and len(gig) == 0:
                            continue                                         # every evalImg is None (:363-365)
                        if npig == 0:
                            continue                                         # :377-378
                        use = kr < max_det
                        ui, us = ki[use], ks[use]
                        sc = dt_scores[ui, us].astype(np.float64)
                        inds = np.argsort(-sc, kind='mergesort')             # :370
                        sorted_scores = sc[inds]
                        dtm = dt_match[ui, a, :, us].T[:, inds]              # [T,D]
                        dig = dt_ignore[ui, a, :, us].T[:, inds]
                        tps = np.logical_and(dtm, np.logical_not(dig))
                        fps = np.logical_and(np.logical_not(dtm), np.logical_not(dig))
                        tp_sum = np.cumsum(tps, axis=1).astype(np.float64)
                        fp_sum = np.cumsum(fps, axis=1).astype(np.float64)
                        for t in range(T):
                            tp, fp = tp_sum[t], fp_sum[t]
                            nd = len(tp)
                            rc = tp / npig
                            pr = tp / (fp + tp + np.spacing(1))
                            recall[t, k, a, m] = rc[-1] if nd else 0
                            pr = np.maximum.accumulate(pr[::-1])[::-1]       # :402-404 precision envelope
                            where = np.searchsorted(rc, p.recThrs, side='left')
                            ok = where < nd                                  # :407-412 (IndexError ends the fill)
                            q = np.zeros(R)
                            ss = np.zeros(R)
                            q[ok] = pr[where[ok]]
                            ss[ok] = sorted_scores[where[ok]]
                            precision[t, :, k, a, m] = q
                            scores[t, :, k, a, m] = ss
        self.eval = dict(params=p, counts=[T, R, K, A, M], precision=precision, recall=recall, scores=scores)
        return self.eval

    def summarize(self):
        """cocoeval.py:426-476 (`_summarizeDets`): the 12 COCO bbox statistics; stats[0] is the
        value evaluate.py:190 returns."""
        if not self.eval:
            raise Exception('Please run accumulate() first')
        p = self.params

        def one(ap=1, iou_thr=None, area='all', max_dets=100):
            a = [i for i, lbl in enumerate(p.areaRngLbl) if lbl == area]
            m = [i for i, md in enumerate(p.maxDets) if md == max_dets]
            s = self.eval['precision'] if ap == 1 else self.eval['recall']
            if iou_thr is not None:
                s = s[np.where(iou_thr == p.iouThrs)[0]]
            s = s[:, :, :, a, m] if ap == 1 else s[:, :, a, m]
            return -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])

        md = p.maxDets
        self.stats = np.array([
            one(1), one(1, iou_thr=.5, max_dets=md[2]), one(1, iou_thr=.75, max_dets=md[2]),
            one(1, area='small', max_dets=md[2]), one(1, area='medium', max_dets=md[2]),
            one(1, area='large', max_dets=md[2]),
            one(0, max_dets=md[0]), one(0, max_dets=md[1]), one(0, max_dets=md[2]),
            one(0, area='small', max_dets=md[2]), one(0, area='medium', max_dets=md[2]),
            one(0, area='large', max_dets=md[2])], dtype=np.float64)
        return self.stats
