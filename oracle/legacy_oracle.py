"""CPU oracle for SURVEY.md 8f row 4: the reference's legacy / offline IoU conventions.
TEST INFRASTRUCTURE ONLY (same rule as odbox_oracle.py: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it).

PARITY UNPINNED.  Restated here, in NumPy float64 like the reference runs them:

* coco.py:169-191  `calc_iou(box, boxes, area)`: boxes in (cx, cy, h, w) layout, +1 pixel on the
  intersection's width / height only (the areas carry no +1), anchor areas handed in;
* coco.py:379-441  the sequential per-GT matcher of `COCO_Iterable.process_image`: GT boxes are
  visited in order, an anchor is (re)assigned when `iou > 0.5` and `iou > max_ious[anchor]`, and
  `max_ious` is a float32 array -- so the running maximum is the ROUNDED IoU and the comparison
  widens it back (an exact tie keeps the first GT only while the IoU is a binary32 value);
* calculate_metrics.py:22-45  `calc_ious(pred_boxes, true_box)`: xyxy, the same +1 convention,
  areas computed from the corners;
* calculate_metrics.py:135-209  `Metric.feed_true_values`: per true box, the best remaining
  prediction by `iou * objectness * class_score` among those with `iou >= 0.35` and
  `objectness >= 0.3` is popped and counted in the confusion matrix; what is left over at the end
  counts as (none, predicted).

The reference leaves the dtype of `np_anchor_boxes` open (only `complete_initialization`, which
nothing calls, sets it; coco.py:486-502): float64 is taken here -- what `np.array` makes of Python
floats and the one dtype for which NumPy's promotion rules give the same result in every NumPy
version.  The reference holds no fixtures for these functions; the restatement is held
by the hand-derived known answers of tests/test_legacy_oracle_kat.py.
"""
import numpy as np

F64 = np.float64


def calc_iou(box, boxes, area):
    """coco.py:169-191.  box (cx, cy, h, w); boxes [N,4] in the same layout; area [N]."""
    box = np.asarray(box, F64)
    boxes = np.asarray(boxes, F64).reshape(-1, 4)
    area = np.asarray(area, F64).reshape(-1)
    half_w, half_h = boxes[:, 3] / 2, boxes[:, 2] / 2
    ax0, ax1 = boxes[:, 0] - half_w, boxes[:, 0] + half_w
    ay0, ay1 = boxes[:, 1] - half_h, boxes[:, 1] + half_h
    x0, x1 = box[0] - box[3] / 2, box[0] + box[3] / 2
    y0, y1 = box[1] - box[2] / 2, box[1] + box[2] / 2
    own_area = box[2] * box[3]
    w = np.maximum(0, np.minimum(x1, ax1) - np.maximum(x0, ax0) + 1)
    h = np.maximum(0, np.minimum(y1, ay1) - np.maximum(y0, ay0) + 1)
    inter = w * h
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / (own_area + area - inter)


def calc_ious(pred_boxes, true_box):
    """calculate_metrics.py:22-45.  xyxy; +1 on the intersection only."""
    p = np.asarray(pred_boxes, F64).reshape(-1, 4)
    t = np.asarray(true_box, F64)
    tarea = (t[2] - t[0]) * (t[3] - t[1])
    parea = (p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 1])
    w = np.maximum(0, np.minimum(t[2], p[:, 2]) - np.maximum(t[0], p[:, 0]) + 1)
    h = np.maximum(0, np.minimum(t[3], p[:, 3]) - np.maximum(t[1], p[:, 1]) + 1)
    inter = w * h
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / (tarea + parea - inter)


def sequential_match(anchor_boxes, anchor_areas, bboxes, labels, threshold=0.5):
    """coco.py:379-441 for one image.  anchor_boxes [N,4] (cx, cy, h, w), bboxes: rows of
    (x0, y0, w, h), labels: the already converted category of each row (`self.cats[cat_id]`).
    Returns (true_bboxes [N,4], true_labels [N] float64, max_ious [N] float32); the caller raises
    ProcessingError when true_labels.sum() == 0 (:439-462)."""
    anchor_boxes = np.asarray(anchor_boxes, F64).reshape(-1, 4)
    n = anchor_boxes.shape[0]
    max_ious = np.zeros(n, np.float32)
    true_bboxes = anchor_boxes.copy()
    true_labels = np.zeros(n)
    for bb, lab in zip(np.asarray(bboxes, F64).reshape(-1, 4), labels):
        if bb[2] <= 3 or bb[3] <= 3:                       # :418-419
            continue
        x0, y0, w, h = bb
        box = np.array([x0 + w / 2, y0 + h / 2, h, w])     # :424-427
        iou = calc_iou(box, anchor_boxes, anchor_areas)
        update = (iou > threshold) & (iou > max_ious)      # :390-392 (float32 widened for the compare)
        true_bboxes[update] = box                          # :404-405
        true_labels[update] = lab
        max_ious[update] = iou[update]                     # :413 (rounds to float32)
    return true_bboxes, true_labels, max_ious


class ConfusionMetric:
    """calculate_metrics.py:65-86,101-212 fed with arrays instead of json records."""

    def __init__(self, num_classes, iou_threshold=0.35, obj_threshold=0.3):
        self.iou_threshold, self.obj_threshold = iou_threshold, obj_threshold
        self.none_id = num_classes                          # :70-73
        self.cm = np.zeros((num_classes + 1, num_classes + 1))
        self.cm_data = []
        self.dexterity = np.zeros((num_classes + 1, 100, num_classes + 1))   # ClassMetric.dexterity per true class

    def feed_true_values(self, true_xywh, true_cats, pred_xyxy, pred_objs, pred_scores, pred_cats):
        pred = [list(map(float, b)) for b in np.asarray(pred_xyxy, F64).reshape(-1, 4)]
        objs, scores, cats = list(map(float, pred_objs)), list(map(float, pred_scores)), list(map(int, pred_cats))
        for bb, tc in zip(np.asarray(true_xywh, F64).reshape(-1, 4), true_cats):
            tc = int(tc)
            if len(pred) == 0:                              # :141-144
                self.cm_data.append((tc, self.none_id))
                self.cm[tc, self.none_id] += 1
                continue
            true_bb = np.array([bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3]])     # :47-53
            ious = calc_ious(np.array(pred), true_bb)
            for k in np.where(ious > self.iou_threshold)[0]:                     # :148-159
                if objs[k] < self.obj_threshold:
                    continue
                self.dexterity[tc, int(scores[k] * 100), cats[k]] += 1
            good = np.where((ious >= self.iou_threshold) & (np.array(objs) >= self.obj_threshold))[0]   # :164
            if len(good) == 0:
                self.cm_data.append((tc, self.none_id))
                self.cm[tc, self.none_id] += 1
                continue
            rank = ious[good] * np.array(objs)[good] * np.array(scores)[good]    # :177
            best = int(good[np.argmax(rank)])
            pc = cats[best]
            for lst in (pred, objs, scores, cats):                               # :184-188
                lst.pop(best)
            self.cm_data.append((tc, pc))
            self.cm[tc, pc] += 1
        for pc in cats:                                                          # :205-209
            self.cm_data.append((self.none_id, pc))
            self.cm[self.none_id, pc] += 1

    def confusion_matrix(self):
        return self.cm.astype(int)

    def per_entry_stats(self):
        """calculate_metrics.py:214-236 without the print: rows of (f1, precision, recall, tp, fp, fn,
        undetected, unlabeled) per class."""
        orig = self.confusion_matrix()
        cm = orig[:-1, :-1]
        sum_pred, sum_true = np.sum(cm, axis=0), np.sum(cm, axis=1)
        rows = []
        for cid in range(self.none_id):
            tp = cm[cid][cid]
            fp, fn = sum_pred[cid] - tp, sum_true[cid] - tp
            rc = tp / (tp + fn + 1e-10) * 100
            pr = tp / (tp + fp + 1e-10) * 100
            f1 = 2 * pr * rc / (pr + rc + 1e-10)
            undetected = orig[cid][self.none_id] / (np.sum(orig[cid, :]) + 1e-10) * 100
            unlabeled = orig[self.none_id][cid] / (np.sum(orig[:, cid]) + 1e-10) * 100
            rows.append((f1, pr, rc, int(tp), int(fp), int(fn), undetected, unlabeled))
        return rows
