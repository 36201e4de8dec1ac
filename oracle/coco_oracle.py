"""CPU oracle for SURVEY.md 8f row 2: detections -> COCO records -> IoU -> greedy GT matching ->
precision/recall.  TEST INFRASTRUCTURE ONLY (same rule as odbox_oracle.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

PARITY UNPINNED.  What is restated here, in plain Python lists / dicts / loops like the reference
runs them:

* evaluate.py:14-55 (`_COCO_result`, `_COCO_gt_annot`), :93-110 (per image: padded GT rows with
  label -1 dropped, padded detection rows NOT dropped), :141-142 (running ids), :172-181 (detection
  area / id / iscrowd);
* cocoeval.py:88-123 (`_prepare`), :167-194 (`computeIoU`), :239-317 (`evaluateImg`), :319-424
  (`accumulate`), :426-497 (`summarize`), :506-515 (`Params`, bbox defaults);
* pycocotools (third party, absent from the reference tree and from this image, version not pinned
  by the reference): `COCO.createIndex` / `getAnnIds` / `loadAnns` ordering (annotations of an
  image in insertion order, images in the order asked) and `maskUtils.iou` on boxes = `bbIou` of
  maskApi.c, restated from its published algorithm:
      for every gt g, every dt d:  w = min(dx+dw, gx+gw) - max(dx, gx);  if w <= 0 -> 0
                                   h = min(dy+dh, gy+gh) - max(dy, gy);  if h <= 0 -> 0
                                   i = w*h;  u = iscrowd[g] ? dw*dh : dw*dh + gw*gh - i;  iou = i/u
  in C doubles (the lists handed over hold Python floats = the fp32 tensor values widened).

It is checked by hand-derived known answers in tests/test_oracle_kat.py; the reference holds no
fixtures for this path.
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------
# evaluate.py: tensors -> COCO dict records
# ----------------------------------------------------------------------------------------

def coco_result(image_id, labels, bboxes, scores):
    """evaluate.py:14-28.  bboxes [R,4] xyxy fp32; w / h are fp32 subtractions, then widened."""
    bb = np.asarray(bboxes, dtype=F32).reshape(-1, 4)
    h = bb[:, 3] - bb[:, 1]
    w = bb[:, 2] - bb[:, 0]
    xywh = np.stack([bb[:, 0], bb[:, 1], w, h]).T.tolist()
    lab = np.asarray(labels).tolist()
    sc = np.asarray(scores, dtype=F32).tolist()
    return [dict(image_id=image_id, category_id=l, bbox=b, score=s) for l, b, s in zip(lab, xywh, sc)]


def coco_gt_annot(image_id, annot_id, image_shape, labels, bboxes):
    """evaluate.py:31-55.  area = b_h * b_w in fp32 (:41), then widened."""
    bb = np.asarray(bboxes, dtype=F32).reshape(-1, 4)
    h = bb[:, 3] - bb[:, 1]
    w = bb[:, 2] - bb[:, 0]
    areas = (h * w).astype(F32).reshape(-1).tolist()
    xywh = np.stack([bb[:, 0], bb[:, 1], w, h]).T.tolist() if len(bb) else []
    lab = np.asarray(labels).tolist()
    image = dict(id=image_id, height=image_shape[0], width=image_shape[1])
    anns = [dict(id=annot_id + k, image_id=image_id, bbox=b, iscrowd=0, area=a, category_id=l)
            for k, (b, a, l) in enumerate(zip(xywh, areas, lab))]
    return image, anns


def build_records(dt_boxes, dt_scores, dt_labels, gt_boxes, gt_labels, image_hw, num_classes):
    """evaluate.py:66-74, 93-110, 141-142, 172-181 for one or more batches already concatenated on
    the image axis.  Returns (gt_dataset dict, list of result annotations)."""
    gt = dict(images=[], annotations=[],
              categories=[dict(supercategory='instance', id=i, name=str(i)) for i in range(num_classes)])
    results = []
    image_id, annot_id = 1, 1
    for b in range(len(dt_boxes)):
        keep = np.asarray(gt_labels[b]) != -1                               # :95-98
        im, anns = coco_gt_annot(image_id, annot_id, image_hw, np.asarray(gt_labels[b])[keep],
                                 np.asarray(gt_boxes[b], dtype=F32)[keep])
        gt['annotations'].extend(anns)
        gt['images'].append(im)
        if len(dt_labels[b]) > 0:                                            # :108
            results.extend(coco_result(image_id, dt_labels[b], dt_boxes[b], dt_scores[b]))
        annot_id += len(anns)
        image_id += 1
    for i, ann in enumerate(results):                                        # :172-179
        bb = ann['bbox']
        ann['area'] = bb[2] * bb[3]
        ann['id'] = i + 1
        ann['iscrowd'] = 0
    return gt, results


# ----------------------------------------------------------------------------------------
# pycocotools: bbIou
# ----------------------------------------------------------------------------------------

def bb_iou(dt, gt, iscrowd):
    """maskUtils.iou(d, g, iscrowd) for box lists: [len(dt), len(gt)] float64 (see module docstring)."""
    out = np.zeros((len(dt), len(gt)), dtype=np.float64)
    for g, G in enumerate(gt):
        ga = float(G[2]) * float(G[3])
        for d, D in enumerate(dt):
            da = float(D[2]) * float(D[3])
            w = min(float(D[2]) + float(D[0]), float(G[2]) + float(G[0])) - max(float(D[0]), float(G[0]))
            if w <= 0:
                continue
            h = min(float(D[3]) + float(D[1]), float(G[3]) + float(G[1])) - max(float(D[1]), float(G[1]))
            if h <= 0:
                continue
            i = w * h
            u = da if iscrowd[g] else da + ga - i
            out[d, g] = i / u
    return out


# ----------------------------------------------------------------------------------------
# cocoeval.py
# ----------------------------------------------------------------------------------------

class Params:
    """cocoeval.py:506-515 (`setDetParams`)."""

    def __init__(self):
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ['all', 'small', 'medium', 'large']


class CocoBoxEval:
    """COCOeval(gt, dt, 'bbox') with useCats = 1, driven from the two record lists."""

    def __init__(self, gt_dataset, results, params=None):
        self.p = params or Params()
        self.img_ids = sorted(im['id'] for im in gt_dataset['images'])       # evaluate.py:185
        self.cat_ids = sorted(c['id'] for c in gt_dataset['categories'])     # cocoeval.py:85
        cats = set(self.cat_ids)
        imgs = set(self.img_ids)
        # _prepare (:88-123): loadAnns(getAnnIds(imgIds, catIds)) keeps, image by image, the insertion
        # order of the annotations and drops categories / images that were not asked for
        self.gts, self.dts = {}, {}
        for ann in gt_dataset['annotations']:
            if ann['category_id'] in cats and ann['image_id'] in imgs:
                ann['ignore'] = bool(ann.get('iscrowd', 0))                  # :111-113
                self.gts.setdefault((ann['image_id'], ann['category_id']), []).append(ann)
        for ann in results:
            if ann['category_id'] in cats and ann['image_id'] in imgs:
                self.dts.setdefault((ann['image_id'], ann['category_id']), []).append(ann)
        self.ious = {}
        self.eval_imgs = []

    # :167-194
    def compute_iou(self, img, cat):
        gt = self.gts.get((img, cat), [])
        dt = self.dts.get((img, cat), [])
        if not gt and not dt:
            return []
        order = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[i] for i in order][:self.p.maxDets[-1]]
        if not dt or not gt:
            return []                                                        # maskUtils.iou on an empty list
        return bb_iou([d['bbox'] for d in dt], [g['bbox'] for g in gt], [int(g['iscrowd']) for g in gt])

    # :239-317
    def evaluate_img(self, img, cat, rng, max_det):
        gt = self.gts.get((img, cat), [])
        dt = self.dts.get((img, cat), [])
        if not gt and not dt:
            return None
        for g in gt:
            g['_ignore'] = 1 if (g['ignore'] or g['area'] < rng[0] or g['area'] > rng[1]) else 0
        gtind = np.argsort([g['_ignore'] for g in gt], kind='mergesort')
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[i] for i in dtind[0:max_det]]
        iscrowd = [int(g['iscrowd']) for g in gt]
        ious = self.ious[img, cat]
        ious = ious[:, gtind] if len(ious) > 0 else ious
        T, G, D = len(self.p.iouThrs), len(gt), len(dt)
        gtm = np.zeros((T, G))
        dtm = np.zeros((T, D))
        gt_ig = np.array([g['_ignore'] for g in gt])
        dt_ig = np.zeros((T, D))
        if len(ious) != 0:
            for ti, t in enumerate(self.p.iouThrs):
                for di, d in enumerate(dt):
                    best = min([t, 1 - 1e-10])
                    m = -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not iscrowd[gi]:
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                            break
                        if ious[di, gi] < best:
                            continue
                        best = ious[di, gi]
                        m = gi
                    if m == -1:
                        continue
                    dt_ig[ti, di] = gt_ig[m]
                    dtm[ti, di] = gt[m]['id']
                    gtm[ti, m] = d['id']
        out = np.array([d['area'] < rng[0] or d['area'] > rng[1] for d in dt]).reshape((1, len(dt)))
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(out, T, 0)))
        return dict(image_id=img, category_id=cat, aRng=rng, maxDet=max_det,
                    dtIds=[d['id'] for d in dt], gtIds=[g['id'] for g in gt], dtMatches=dtm, gtMatches=gtm,
                    dtScores=[d['score'] for d in dt], gtIgnore=gt_ig, dtIgnore=dt_ig)

    # :125-165
    def evaluate(self):
        self.ious = {(i, c): self.compute_iou(i, c) for i in self.img_ids for c in self.cat_ids}
        md = self.p.maxDets[-1]
        self.eval_imgs = [self.evaluate_img(i, c, rng, md)
                          for c in self.cat_ids for rng in self.p.areaRng for i in self.img_ids]
        return self.eval_imgs

    # :319-424
    def accumulate(self):
        p = self.p
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(self.cat_ids), len(p.areaRng), len(p.maxDets)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        scores = -np.ones((T, R, K, A, M))
        I = len(self.img_ids)
        for k in range(K):
            for a in range(A):
                for m, max_det in enumerate(p.maxDets):
                    E = [self.eval_imgs[k * A * I + a * I + i] for i in range(I)]
                    E = [e for e in E if e is not None]
                    if not E:
                        continue
                    dt_scores = np.concatenate([e['dtScores'][0:max_det] for e in E])
                    inds = np.argsort(-dt_scores, kind='mergesort')
                    sorted_scores = dt_scores[inds]
                    dtm = np.concatenate([e['dtMatches'][:, 0:max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e['dtIgnore'][:, 0:max_det] for e in E], axis=1)[:, inds]
                    gt_ig = np.concatenate([e['gtIgnore'] for e in E])
                    npig = np.count_nonzero(gt_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum = np.cumsum(tps, axis=1).astype(np.float64)
                    fp_sum = np.cumsum(fps, axis=1).astype(np.float64)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = (tp / (fp + tp + np.spacing(1))).tolist()
                        q = [0.0] * R
                        ss = [0.0] * R
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        where = np.searchsorted(rc, p.recThrs, side='left')
                        for ri, pi in enumerate(where):
                            if pi >= nd:
                                break                                        # the reference's try/except
                            q[ri] = pr[pi]
                            ss[ri] = sorted_scores[pi]
                        precision[t, :, k, a, m] = np.array(q)
                        scores[t, :, k, a, m] = np.array(ss)
        self.eval = dict(precision=precision, recall=recall, scores=scores, counts=[T, R, K, A, M])
        return self.eval

    # :426-476
    def summarize(self):
        p = self.p

        def one(ap, iou_thr=None, area='all', max_dets=100):
            a = [i for i, lbl in enumerate(p.areaRngLbl) if lbl == area]
            m = [i for i, md in enumerate(p.maxDets) if md == max_dets]
            s = self.eval['precision'] if ap else self.eval['recall']
            if iou_thr is not None:
                s = s[np.where(iou_thr == p.iouThrs)[0]]
            s = s[:, :, :, a, m] if ap else s[:, :, a, m]
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


def dense_eval(dt_boxes, dt_scores, dt_labels, gt_boxes, gt_labels, image_hw, num_classes, params=None):
    """Run the restated pipeline and lay the per-(image, category, area range) results out as the
    dense tensors the device path produces (include/odbox.h odbox_coco_match):

      dt_match  [B,A,T,R] int32  1 + GT row (in the image's padded GT tensor) matched, 0 = none
      dt_ignore [B,A,T,R] uint8
      gt_match  [B,A,T,G] int32  1 + detection slot matched, 0 = none
      gt_ignore [B,A,G]   uint8  0 counts / 1 ignored / 255 row absent (padding, unknown category)
      dt_rank   [B,R]     int32  position in its (image, category) list by (score desc, slot asc),
                                 -1 = not evaluated (unknown category or beyond maxDets[-1])
    plus the CocoBoxEval object (for .ious, accumulate, summarize)."""
    p = params or Params()
    dt_labels = np.asarray(dt_labels)
    gt_labels = np.asarray(gt_labels)
    B, R = dt_labels.shape
    G = gt_labels.shape[1]
    A, T = len(p.areaRng), len(p.iouThrs)
    gt_ds, results = build_records(dt_boxes, dt_scores, dt_labels, gt_boxes, gt_labels, image_hw, num_classes)
    ev = CocoBoxEval(gt_ds, results, p)
    ev.evaluate()
    # id -> (image, row/slot): GT ids run over the kept rows in order, detection ids over all slots
    gt_row = {}
    nxt = 1
    for b in range(B):
        for g in range(G):
            if gt_labels[b, g] != -1:
                gt_row[nxt] = g
                nxt += 1
    dt_match = np.zeros((B, A, T, R), np.int32)
    dt_ignore = np.zeros((B, A, T, R), np.uint8)
    gt_match = np.zeros((B, A, T, G), np.int32)
    gt_ignore = np.full((B, A, G), 255, np.uint8)
    dt_rank = np.full((B, R), -1, np.int32)
    I, K = B, num_classes
    for k in range(K):
        for a in range(A):
            for i in range(I):
                e = ev.eval_imgs[k * A * I + a * I + i]
                if e is None:
                    continue
                for j, did in enumerate(e['dtIds']):
                    slot = (did - 1) - i * R
                    dt_rank[i, slot] = j
                    for t in range(T):
                        m = int(e['dtMatches'][t, j])
                        dt_match[i, a, t, slot] = 0 if m == 0 else gt_row[m] + 1
                        dt_ignore[i, a, t, slot] = 1 if e['dtIgnore'][t, j] else 0
                for j, gid in enumerate(e['gtIds']):
                    g = gt_row[gid]
                    gt_ignore[i, a, g] = int(e['gtIgnore'][j])
                    for t in range(T):
                        m = int(e['gtMatches'][t, j])
                        gt_match[i, a, t, g] = 0 if m == 0 else (m - 1) - i * R + 1
    return dict(dt_match=dt_match, dt_ignore=dt_ignore, gt_match=gt_match, gt_ignore=gt_ignore,
                dt_rank=dt_rank, evaluator=ev)
