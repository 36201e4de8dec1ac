"""Drop-in for the reference's offline confusion-matrix report (calculate_metrics.py) on MI355X
(SURVEY.md §8f row 4).  `calc_ious` (calculate_metrics.py:22-45) runs on the GPU in float64 (HIP
only, no CPU fallback); `Metric.feed_true_values` (:101-209) asks for the whole true x predicted IoU
table of an image in ONE launch and then does the reference's own sequential bookkeeping (pop the
best prediction per true box) on the host over that table -- popping a prediction removes a column,
it never changes another column's value, so the table is computed once.

`Metric` takes the predictions either as the path of the results json (like the reference) or as the
already loaded list of {'filename', 'annotations'} records.
"""
import json

import numpy as np
import torch

from . import _lib


def calc_ious(pred_boxes, true_box):
    """calculate_metrics.py:22-45: pred_boxes [P,4] against true_box [4], xyxy, +1 pixel on the
    intersection only.  true_box may also be [T,4]: T rows of results."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    t = _lib.as_dev(true_box, torch.float64, dev)
    single = t.dim() == 1
    t = t.reshape(-1, 4)
    p = _lib.as_dev(pred_boxes, torch.float64, dev).reshape(-1, 4)
    out = torch.empty((t.shape[0], p.shape[0]), dtype=torch.float64, device=dev)
    _lib.check(lib.odbox_legacy_iou(_lib.ptr(t), _lib.ptr(p), None, t.shape[0], p.shape[0], 1, _lib.ptr(out),
                                    _lib.stream_ptr()), 'odbox_legacy_iou')
    return out[0] if single else out


def convert_true_bbox_to_xyxy(true_bb):
    """calculate_metrics.py:47-53."""
    x0, y0, w, h = true_bb
    return np.array([x0, y0, x0 + w, y0 + h])


class ClassMetric:
    """calculate_metrics.py:55-63."""

    def __init__(self, cid, cname, cat_ids):
        self.cname = cname
        self.cid = cid
        self.dexterity = np.zeros((100, len(cat_ids)))

    def feed_dexterity(self, score, cid):
        self.dexterity[int(score * 100), cid] += 1


class Metric:
    """calculate_metrics.py:65-236."""

    def __init__(self, prediction_json, cat_ids, cat_names):
        self.iou_threshold = 0.35
        self.obj_threshold = 0.3
        self.none_id = len(cat_ids)
        self.none_name = 'none'
        cat_ids.append(self.none_id)                    # the reference extends the caller's lists too (:72-73)
        cat_names.append(self.none_name)
        self.cat_ids = cat_ids
        self.cat_names = cat_names
        self.cm_data = []
        self.cm = np.zeros(shape=[len(self.cat_ids), len(self.cat_ids)])
        self.cmetrics = {cid: ClassMetric(cid, cname, self.cat_ids) for cid, cname in zip(cat_ids, cat_names)}
        self.fn2ann = self.open_predictions(prediction_json)

    def open_predictions(self, input_fn):
        if isinstance(input_fn, (str, bytes)):
            with open(input_fn, 'r') as fin:
                results = json.load(fin)
        else:
            results = input_fn
        return {js['filename']: js['annotations'] for js in results}

    def feed_true_values(self, filename, true_anns):
        pred_anns = self.fn2ann[filename]
        alive = list(range(len(pred_anns)))             # indices of the predictions not popped yet
        objs = np.array([a['objectness'] for a in pred_anns], dtype=np.float64)
        scores = np.array([a['class_score'] for a in pred_anns], dtype=np.float64)
        table = None
        if len(pred_anns) and len(true_anns):
            true_xyxy = np.stack([convert_true_bbox_to_xyxy(a['bbox']) for a in true_anns])
            table = calc_ious(np.array([a['bbox'] for a in pred_anns]), true_xyxy).cpu().numpy()
        for t, true_ann in enumerate(true_anns):
            true_cat_id = true_ann['category_id']
            if len(alive) == 0:                         # :141-144
                self.cm_data.append((true_cat_id, self.none_id))
                self.cm[true_cat_id, self.none_id] += 1
                continue
            idx = np.array(alive)
            ious = table[t, idx]
            for k in np.where(ious > self.iou_threshold)[0]:                      # :148-159
                ann = pred_anns[idx[k]]
                if ann['objectness'] < self.obj_threshold:
                    continue
                self.cmetrics[true_cat_id].feed_dexterity(ann['class_score'], ann['category_id'])
            good = np.where((ious >= self.iou_threshold) & (objs[idx] >= self.obj_threshold))[0]   # :164
            if len(good) == 0:
                self.cm_data.append((true_cat_id, self.none_id))
                self.cm[true_cat_id, self.none_id] += 1
                continue
            rank = ious[good] * objs[idx][good] * scores[idx][good]               # :172-177
            best = good[np.argmax(rank)]
            pred_ann = pred_anns[alive.pop(int(best))]                            # :184-188
            self.cm_data.append((true_cat_id, pred_ann['category_id']))
            self.cm[true_cat_id, pred_ann['category_id']] += 1
        for k in alive:                                                           # :205-209
            pred_cat_id = pred_anns[k]['category_id']
            self.cm_data.append((self.none_id, pred_cat_id))
            self.cm[self.none_id, pred_cat_id] += 1

    def confusion_matrix(self):
        return self.cm.astype(int)

    def per_entry_stats(self, report=True):
        """calculate_metrics.py:214-236, all classes at once: one row (f1, precision, recall, tp, fp, fn, undetected,
        unlabeled) per real class, in percent where the reference prints percent, with the reference's `+ 1e-10`
        guards and its order of operations (float64, so the rows equal the per-class loop bit for bit).  The report
        (one line per class) goes through `format_stats`; report=False only returns the rows."""
        full = self.confusion_matrix()
        k = self.none_id                                    # the trailing 'none' class is not a row
        inner = full[:k, :k]
        tp = np.diagonal(inner)
        fp = inner.sum(axis=0) - tp
        fn = inner.sum(axis=1) - tp
        recall = tp / (tp + fn + 1e-10) * 100
        precision = tp / (tp + fp + 1e-10) * 100
        f1 = 2 * precision * recall / (precision + recall + 1e-10)
        undetected = full[:k, k] / (full[:k, :].sum(axis=1) + 1e-10) * 100
        unlabeled = full[k, :k] / (full[:, :k].sum(axis=0) + 1e-10) * 100
        rows = [(f1[c], precision[c], recall[c], int(tp[c]), int(fp[c]), int(fn[c]), undetected[c], unlabeled[c])
                for c in range(k)]
        if report:
            print(format_stats(self.cat_names[:k], rows))
        return rows


def format_stats(names, rows):
    """The per-class report of `Metric.per_entry_stats` as text: the reference's eight figures per class."""
    head = ('class', 'f1', 'precision', 'recall', 'tp', 'fp', 'fn', 'undetected', 'unlabelled')
    lines = ['%20s %6s %9s %6s %5s %5s %5s %10s %10s' % head]
    for cid, (name, r) in enumerate(zip(names, rows)):
        lines.append('%17s/%2d %6.1f %9.1f %6.1f %5d %5d %5d %10.1f %10.1f' % ((name, cid) + tuple(r)))
    return '\n'.join(lines)
