"""Hand-derived known answers for the evaluation-side oracle (oracle/coco_oracle.py): COCO records,
pycocotools' bbIou, cocoeval.py's greedy matching and the precision/recall summary.  The reference
holds no fixtures for this path: parity unpinned, these cases are what pins it."""
import numpy as np

from oracle import coco_oracle as CO

HW = (512, 512)


def run(dt_boxes, dt_scores, dt_labels, gt_boxes, gt_labels, C=3):
    f = lambda x, d: np.asarray(x, dtype=d)[None]
    return CO.dense_eval(f(dt_boxes, np.float32), f(dt_scores, np.float32), f(dt_labels, np.int32),
                         f(gt_boxes, np.float32), f(gt_labels, np.int32), HW, C)


def test_records_follow_evaluate_py():
    res = CO.coco_result(7, [2], [[1.5, 2.0, 11.5, 22.0]], [0.75])
    assert res == [dict(image_id=7, category_id=2, bbox=[1.5, 2.0, 10.0, 20.0], score=0.75)]
    im, anns = CO.coco_gt_annot(3, 40, HW, [1, 0], [[0, 0, 10, 5], [2, 2, 4, 8]])
    assert im == dict(id=3, height=512, width=512)
    assert [a['id'] for a in anns] == [40, 41] and [a['area'] for a in anns] == [50.0, 12.0]
    assert anns[1]['bbox'] == [2.0, 2.0, 2.0, 6.0] and anns[0]['iscrowd'] == 0


def test_bb_iou_known_values():
    iou = CO.bb_iou([[0, 0, 10, 10], [20, 20, 5, 5], [0, 0, 0, 0]], [[5, 0, 10, 10], [0, 0, 10, 10]], [0, 0])
    assert iou[0, 0] == 50.0 / 150.0 and iou[0, 1] == 1.0
    assert (iou[1] == 0).all() and (iou[2] == 0).all()                 # disjoint; zero-size padding row
    assert CO.bb_iou([[0, 0, 10, 10]], [[10, 0, 10, 10]], [0])[0, 0] == 0.0    # touching edges: w == 0
    assert CO.bb_iou([[0, 0, 10, 10]], [[5, 0, 10, 10]], [1])[0, 0] == 0.5     # crowd: inter / det area


def test_higher_score_takes_the_gt_and_thresholds_cut():
    # two detections on one GT (IoU 1.0 and 0.81): the better-scored one matches at every threshold it
    # reaches, the other stays unmatched even where it would pass
    r = run([[0, 0, 9, 10], [0, 0, 10, 10]], [0.9, 0.8], [1, 1], [[0, 0, 10, 10]], [1])
    assert r['dt_rank'].tolist() == [[0, 1]]
    iou0 = 90.0 / 100.0
    thr = CO.Params().iouThrs
    for t in range(10):
        hit = iou0 >= min(thr[t], 1 - 1e-10)
        assert r['dt_match'][0, 0, t].tolist() == ([1, 0] if hit else [0, 1])
        assert r['gt_match'][0, 0, t].tolist() == ([1] if hit else [2])


def test_area_ranges_ignore_gt_and_unmatched_detections():
    # GT 20x20 = 400 px: counted in 'all' and 'small', ignored in 'medium' / 'large'
    r = run([[0, 0, 20, 20], [100, 100, 200, 200]], [0.9, 0.5], [0, 0], [[0, 0, 20, 20]], [0], C=1)
    assert r['gt_ignore'][0, :, 0].tolist() == [0, 0, 1, 1]
    assert r['dt_match'][0, :, 0, 0].tolist() == [1, 1, 1, 1]           # an ignored GT still matches ...
    assert r['dt_ignore'][0, :, 0, 0].tolist() == [0, 0, 1, 1]          # ... and passes its flag on
    # the unmatched 100x100 = 10^4 px detection is ignored outside its own range ('small', 'medium')
    assert r['dt_ignore'][0, :, 0, 1].tolist() == [0, 1, 1, 0]


def test_equal_iou_goes_to_the_later_gt_and_counted_gt_wins_over_ignored():
    # identical GT rows 0 and 1: the loop keeps the LAST row that is >= the best so far
    r = run([[0, 0, 10, 10]], [0.9], [0], [[0, 0, 10, 10], [0, 0, 10, 10]], [0, 0], C=1)
    assert r['dt_match'][0, 0, 0].tolist() == [2]
    # rows: 0 = small exact copy (ignored in 'large'), 1 = large loose box (counted in 'large')
    r = run([[0, 0, 100, 100]], [0.9], [0], [[0, 0, 100, 100], [0, 0, 100, 120]], [0, 0], C=1)
    assert r['dt_match'][0, 0, 0].tolist() == [1]                       # 'all': best IoU
    assert r['gt_ignore'][0, 3].tolist() == [0, 0]
    r = run([[0, 0, 90, 90]], [0.9], [0], [[0, 0, 90, 90], [0, 0, 100, 100]], [0, 0], C=1)
    assert r['gt_ignore'][0, 3].tolist() == [1, 0]                      # 8100 < 96^2 <= 10^4
    assert r['dt_match'][0, 3, 0].tolist() == [2]                       # IoU .81 on the counted row beats 1.0 on the ignored one
    assert r['dt_match'][0, 0, 0].tolist() == [1]


def test_categories_padding_and_truncation():
    # different category never matches; padded GT (-1) is absent; zero detection rows are class-0 detections
    r = run([[0, 0, 10, 10], [0, 0, 0, 0]], [0.9, 0.0], [2, 0], [[0, 0, 10, 10], [-1, -1, -1, -1]], [1, -1])
    assert r['dt_match'].max() == 0 and r['gt_ignore'][0, 0].tolist() == [0, 255]
    assert r['dt_rank'].tolist() == [[0, 0]]
    # a label outside [0, C) is not evaluated at all
    r = run([[0, 0, 10, 10]], [0.9], [5], [[0, 0, 10, 10]], [5])
    assert r['dt_rank'].tolist() == [[-1]] and r['gt_ignore'][0, 0].tolist() == [255]
    # more than maxDets[-1] = 100 detections of one category: the 100 best stay, ties by slot
    n = 103
    sc = np.full(n, 0.5, np.float32)
    sc[50] = 0.9
    r = run(np.tile([[0, 0, 10, 10]], (n, 1)), sc, np.zeros(n, int), [[0, 0, 10, 10]], [0], C=1)
    rank = r['dt_rank'][0]
    assert rank[50] == 0 and rank[0] == 1 and rank[49] == 50 and rank[51] == 51
    assert rank[99] == 99 and (rank[100:] == -1).all()
    assert r['dt_match'][0, 0, 0, 50] == 1 and r['dt_match'][0, 0, 0].sum() == 1


def test_summary_of_a_perfect_and_of_an_empty_detector():
    gt_b = [[10, 10, 40, 40], [100, 100, 220, 220], [300, 300, 360, 380]]
    r = run(gt_b, [0.9, 0.8, 0.7], [0, 1, 1], gt_b, [0, 1, 1], C=2)
    ev = r['evaluator']
    ev.accumulate()
    s = ev.summarize()
    # precision is tp / (tp + fp + eps): 1 - 2.2e-16 for a perfect detector
    np.testing.assert_allclose(s[[0, 1, 2, 8]], 1.0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(s[[3, 4, 5]], 1.0, rtol=0, atol=1e-12)   # 900 small, 4800 medium, 14400 large
    assert s[6] == 0.75                                                # maxDets 1: class 1 finds one of its two GT
    r = run([[0, 0, 0, 0]], [0.0], [0], gt_b, [0, 1, 1], C=2)
    ev = r['evaluator']
    ev.accumulate()
    s = ev.summarize()
    assert s[0] == 0.0 and s[8] == 0.0
