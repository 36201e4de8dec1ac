"""Hand-derived known answers for oracle/legacy_oracle.py (coco.py:169-191,
:379-441 and calculate_metrics.py:22-45, :135-209).  The reference holds no fixtures for these
functions, so these cases -- every expected value worked out by hand from the reference's formulas,
as exact fractions -- are what pins the restatement (parity unpinned otherwise)."""
import numpy as np

from oracle import legacy_oracle as L


def test_calc_iou_plus_one_pixel_convention_and_layout():
    box = [10, 10, 10, 10]                                  # (cx, cy, h, w): x 5..15, y 5..15, area 100
    anchors = [[10, 10, 10, 10],                            # the same box: inter 11*11, areas without the +1
               [20, 10, 10, 10],                            # touching edge at x = 15: w = 15-15+1 = 1, h = 11
               [100, 100, 10, 10],                          # far away
               [21, 10, 10, 10],                            # one pixel apart: w = 15-16+1 = 0
               [20.5, 10, 10, 10]]                          # half a pixel apart: w = 0.5
    got = L.calc_iou(box, anchors, [100] * 5)
    np.testing.assert_array_equal(got, [121 / 79, 11 / 189, 0.0, 0.0, 5.5 / 194.5])
    assert got[0] > 1.0                                     # the convention's quirk: IoU above 1
    # the anchor area is an INPUT (np_anchor_areas), not recomputed
    assert L.calc_iou(box, anchors[:1], [50])[0] == 121 / 29
    # layout: index 2 is the height, index 3 the width.  box x 6..14, y 18..22; anchor x 11..13, y 18..24
    assert L.calc_iou([10, 20, 4, 8], [[12, 21, 6, 2]], [12])[0] == 15 / 29       # (swapped layout: 12/32)


def test_calc_ious_xyxy():
    preds = [[0, 0, 10, 10], [5, 5, 15, 15], [11, 0, 20, 10], [10, 0, 20, 10]]
    got = L.calc_ious(preds, [0, 0, 10, 10])
    # identical: 11*11 / (100+100-121); shifted: 6*6 / (200-36); one pixel apart: 0; touching: 1*11 / (200-11)
    np.testing.assert_array_equal(got, [121 / 79, 36 / 164, 0.0, 11 / 189])


def test_sequential_matcher_threshold_is_strict_and_small_boxes_are_skipped():
    anchors = [[10, 10, 9, 9], [10, 10, 9, 9], [50, 50, 10, 10]]         # x 5.5..14.5: inter with the GT = 10*10
    gt = [[5, 5, 10, 10]]                                                # -> box (10, 10, 10, 10)
    tb, tl, mi = L.sequential_match(anchors, [200, 199, 100], gt, [7])
    # anchor 0: 100 / (100+200-100) = 0.5 exactly -> not positive; anchor 1: 100/199 > 0.5; anchor 2: 0
    np.testing.assert_array_equal(tl, [0, 7, 0])
    np.testing.assert_array_equal(tb, [[10, 10, 9, 9], [10, 10, 10, 10], [50, 50, 10, 10]])
    np.testing.assert_array_equal(mi, np.array([0, 100 / 199, 0], np.float32))
    # w <= 3 or h <= 3 rows are skipped (coco.py:418-419), whatever they overlap
    tb, tl, _ = L.sequential_match([[10, 10, 3, 10]], [1], [[5, 8.5, 10, 3]], [4])
    assert tl[0] == 0 and (tb == [[10, 10, 3, 10]]).all()


def test_sequential_matcher_ties_go_to_the_first_gt_only_while_the_iou_is_a_binary32_value():
    gt = [[5, 5, 10, 10], [5, 5, 10, 10]]                                # the same box twice, labels 1 then 2
    anchors = [[10, 10, 9, 9]]                                           # inter 100
    # 100/128 is a binary32 value: the second GT's iou is not > max_ious -> label 1 stays
    assert L.sequential_match(anchors, [128], gt, [1, 2])[1][0] == 1
    # 100/150 rounds UP to float32: the second GT (same float64 iou) is below the stored maximum
    assert L.sequential_match(anchors, [150], gt, [1, 2])[1][0] == 1
    # 121/79 rounds DOWN to float32 (max_ious is a float32 array, coco.py:379): the same iou now
    # exceeds the stored maximum and the SECOND GT takes the anchor over
    assert L.sequential_match([[10, 10, 10, 10]], [100], gt, [1, 2])[1][0] == 2
    # a later GT with a larger iou always takes over, a smaller one never does
    anchors = [[10, 10, 10, 10]]
    big, small = [5, 5, 10, 10], [5, 5, 10, 8]                           # iou 121/79 and 11*9/(80+100-99) = 99/81
    tb, tl, mi = L.sequential_match(anchors, [100], [small, big], [1, 2])
    assert tl[0] == 2 and (tb[0] == [10, 10, 10, 10]).all() and mi[0] == np.float32(121 / 79)
    tb, tl, mi = L.sequential_match(anchors, [100], [big, small], [1, 2])
    assert tl[0] == 1 and mi[0] == np.float32(121 / 79)


def test_confusion_matrix_matching():
    m = L.ConfusionMetric(num_classes=3)
    true_xywh = [[0, 0, 10, 10], [100, 100, 10, 10], [200, 200, 10, 10]]
    true_cats = [0, 1, 2]
    pred = [[0, 0, 10, 10],          # on true 0: iou 121/79
            [1, 1, 11, 11],          # on true 0 too: 10*10/(200-100) = 1.0, lower objectness * score
            [100, 100, 110, 110],    # on true 1, but objectness below 0.3 -> cannot be matched
            [300, 300, 310, 310]]    # on nothing -> (none, its class)
    objs = [0.9, 0.5, 0.2, 0.8]
    scores = [0.8, 0.9, 0.9, 0.7]
    cats = [0, 1, 1, 2]
    m.feed_true_values(true_xywh, true_cats, pred, objs, scores, cats)
    # true 0: ranks 121/79*0.9*0.8 = 1.10 vs 1.0*0.5*0.9 = 0.45 -> prediction 0 (class 0) popped
    # true 1: only prediction 2 overlaps and its objectness is too low -> (1, none)
    # true 2: nothing overlaps -> (2, none); left over: predictions 1, 2, 3 -> (none, 1), (none, 1), (none, 2)
    assert m.cm_data == [(0, 0), (1, 3), (2, 3), (3, 1), (3, 1), (3, 2)]
    want = np.zeros((4, 4), int)
    for t, p in m.cm_data:
        want[t, p] += 1
    np.testing.assert_array_equal(m.confusion_matrix(), want)
    # dexterity: true 0 saw prediction 0 (score 0.8 -> bin 80, class 0) and prediction 1 (bin 90, class 1)
    assert m.dexterity[0, 80, 0] == 1 and m.dexterity[0, 90, 1] == 1 and m.dexterity.sum() == 2
    f1, pr, rc, tp, fp, fn, undetected, unlabeled = m.per_entry_stats()[0]
    assert (tp, fp, fn) == (1, 0, 0) and abs(pr - 100) < 1e-6 and abs(rc - 100) < 1e-6
    assert m.per_entry_stats()[1][3:6] == (0, 0, 0) and abs(m.per_entry_stats()[1][6] - 100) < 1e-6


def test_confusion_matrix_with_no_predictions():
    m = L.ConfusionMetric(num_classes=2)
    m.feed_true_values([[0, 0, 5, 5]], [1], np.zeros((0, 4)), [], [], [])
    assert m.cm_data == [(1, 2)]
