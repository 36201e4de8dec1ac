"""GPU parity of the evaluation side (evaluate.py:14-55,93-110,172-181,
cocoeval.py:167-194,239-317 and the accumulate/summarize bookkeeping over it) against
oracle/coco_oracle.py.  Bar: match tables, ignore flags and ranks bit-exact; IoUs bit-exact (fp64,
same association as pycocotools' bbIou); precision / recall / stats equal as doubles."""
import numpy as np
import pytest
import torch

from object_detection_amd import cocoeval as CE
from object_detection_amd import workloads
from oracle import coco_oracle as CO

pytestmark = pytest.mark.gpu
HW = (512, 512)


def _check(d, C, params=None):
    ref = CO.dense_eval(d['dt_boxes'], d['dt_scores'], d['dt_labels'], d['gt_boxes'], d['gt_labels'], HW, C, params)
    out = CE.evaluate_img(d['dt_boxes'], d['dt_scores'], d['dt_labels'], d['gt_boxes'], d['gt_labels'], C, params)
    torch.cuda.synchronize()
    for k in ('dt_rank', 'gt_ignore', 'dt_match', 'dt_ignore', 'gt_match'):
        np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k], err_msg=k)
    return ref, out


@pytest.mark.parametrize('kw', [
    dict(B=6, R=100, G=64, C=80),
    dict(B=4, R=100, G=64, C=80, few_classes=3),            # many GT / detections per category: ties, duplicates
    dict(B=3, R=40, G=20, C=5, valid_gt=20),                # no padded GT rows
    dict(B=3, R=100, G=130, C=2, valid_gt=130, S=1024),     # G > 64: several GT rows per lane
    dict(B=2, R=100, G=300, C=1, valid_gt=290),             # crowd-like, one class
    dict(B=2, R=7, G=3, C=4, valid_gt=0),                   # images without GT and without detections
])
def test_match_tables_equal_oracle(kw):
    C = kw['C']
    ref, _ = _check(workloads.eval_inputs(seed=2000 + kw['B'] * 7 + kw['G'], **kw), C)
    if kw.get('valid_gt', 20):
        assert (ref['dt_match'] > 0).any()


def test_more_detections_than_max_dets_and_unknown_labels():
    d = workloads.eval_inputs(B=2, R=260, G=64, C=2, few_classes=2, valid_gt=40, seed=11)
    d['dt_labels'][0, ::17] = 9                                  # category the evaluator does not know
    d['dt_labels'][1, 5] = -3
    d['gt_labels'][0, 3] = 7
    ref, out = _check(d, 2)
    assert (ref['dt_rank'] == -1).any() and ref['dt_rank'].max() == 99
    assert (ref['gt_ignore'][0, :, 3] == 255).all()


def test_exact_duplicates_and_equal_scores():
    d = workloads.eval_inputs(B=2, R=60, G=16, C=1, valid_gt=8, seed=5)
    d['gt_boxes'][:, 1] = d['gt_boxes'][:, 0]                    # identical GT rows: tie goes to the later one
    d['dt_boxes'][:, 0] = d['gt_boxes'][:, 0]
    d['dt_boxes'][:, 1] = d['gt_boxes'][:, 0]
    d['dt_scores'][:, :6] = 0.75                                 # equal scores: slot order
    _check(d, 1)


def test_custom_thresholds_and_ranges():
    p = CO.Params()
    p.iouThrs = np.array([0.3, 0.5, 1.0])
    p.areaRng = [[0, 1e10], [50 ** 2, 200 ** 2]]
    p.areaRngLbl = ['all', 'mid']
    q = CE.Params()
    q.iouThrs, q.areaRng, q.areaRngLbl = p.iouThrs, p.areaRng, p.areaRngLbl
    d = workloads.eval_inputs(B=3, R=50, G=32, C=3, seed=21)
    ref = CO.dense_eval(d['dt_boxes'], d['dt_scores'], d['dt_labels'], d['gt_boxes'], d['gt_labels'], HW, 3, p)
    out = CE.evaluate_img(d['dt_boxes'], d['dt_scores'], d['dt_labels'], d['gt_boxes'], d['gt_labels'], 3, q)
    for k in ('dt_rank', 'gt_ignore', 'dt_match', 'dt_ignore', 'gt_match'):
        np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k], err_msg=k)


def test_iou_matrix_and_records_are_bit_exact():
    d = workloads.eval_inputs(B=3, R=30, G=16, C=4, seed=3)
    iou = CE.compute_iou(d['dt_boxes'], d['gt_boxes']).cpu().numpy()
    for b in range(3):
        dt = CO.coco_result(1, d['dt_labels'][b], d['dt_boxes'][b], d['dt_scores'][b])
        _, gt = CO.coco_gt_annot(1, 1, HW, d['gt_labels'][b], d['gt_boxes'][b])
        ref = CO.bb_iou([x['bbox'] for x in dt], [x['bbox'] for x in gt], [0] * len(gt))
        np.testing.assert_array_equal(iou[b], ref)
    xywh, area = CE.coco_records(d['gt_boxes'])
    _, gt = CO.coco_gt_annot(1, 1, HW, d['gt_labels'][0], d['gt_boxes'][0])
    np.testing.assert_array_equal(xywh[0].cpu().numpy(), np.array([g['bbox'] for g in gt], np.float32))
    np.testing.assert_array_equal(area[0].cpu().numpy(), np.array([g['area'] for g in gt], np.float32))
    _, a64 = CE.coco_records(d['dt_boxes'], want_area_f64=True)
    _, res = CO.build_records(d['dt_boxes'], d['dt_scores'], d['dt_labels'], d['gt_boxes'], d['gt_labels'], HW, 4)
    np.testing.assert_array_equal(a64.cpu().numpy().reshape(-1), np.array([r['area'] for r in res]))


@pytest.mark.parametrize('few', [None, 4])
def test_accumulate_and_summarize_equal_oracle(few):
    C = 12
    batches = [workloads.eval_inputs(B=5, R=100, G=64, C=C, few_classes=few, seed=70 + i) for i in range(2)]
    ev = CE.COCOeval(C)
    for d in batches:
        ev.add_batch(d['dt_boxes'], d['dt_scores'], d['dt_labels'], d['gt_boxes'], d['gt_labels'])
    ev.accumulate()
    stats = ev.summarize()
    cat = {k: np.concatenate([d[k] for d in batches]) for k in batches[0]}
    ref = CO.dense_eval(cat['dt_boxes'], cat['dt_scores'], cat['dt_labels'], cat['gt_boxes'], cat['gt_labels'], HW, C)
    rev = ref['evaluator']
    rev.accumulate()
    for k in ('precision', 'recall', 'scores'):
        np.testing.assert_array_equal(ev.eval[k], rev.eval[k], err_msg=k)
    np.testing.assert_array_equal(stats, rev.summarize())
    assert 0 < stats[0] < 1


def test_empty_and_bad_arguments():
    z = torch.zeros
    out = CE.evaluate_img(z(2, 5, 4), z(2, 5), z(2, 5, dtype=torch.int32), z(2, 0, 4), z(2, 0, dtype=torch.int32), 3)
    assert out['dt_match'].shape == (2, 4, 10, 5) and int(out['dt_match'].abs().sum()) == 0
    assert out['dt_rank'].cpu().tolist() == [[0, 1, 2, 3, 4]] * 2          # zero rows = category-0 detections
    with pytest.raises(ValueError):
        CE.evaluate_img(z(2, 5, 4), z(2, 4), z(2, 5), z(2, 1, 4), z(2, 1), 3)
