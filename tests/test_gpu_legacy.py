"""SURVEY.md 8f row 4 on the GPU: the legacy +1-pixel IoU conventions (coco.py:169-191,
calculate_metrics.py:22-45), the sequential per-GT matcher (coco.py:379-441) and the confusion-matrix
matching (calculate_metrics.py:101-209) against oracle/legacy_oracle.py.  float64 work: bit-exact."""
import numpy as np
import pytest
import torch

from object_detection_amd import calculate_metrics, coco
from oracle import legacy_oracle as L

pytestmark = pytest.mark.gpu


def cxcyhw_anchors(rng, n, size=512.0):
    c = rng.uniform(0, size, (n, 2))
    hw = rng.uniform(4, 160, (n, 2))
    return np.concatenate([c, hw], axis=1)


def gt_xywh(rng, b, g, size=512.0, pad_from=None):
    xy = rng.uniform(0, size * 0.8, (b, g, 2))
    wh = rng.uniform(2, 200, (b, g, 2))                     # some rows fall under the w <= 3 / h <= 3 filter
    gt = np.concatenate([xy, wh], axis=2)
    if pad_from is not None:
        gt[:, pad_from:] = -1.0
    return gt


def test_kats_through_the_kernels():
    box = [10, 10, 10, 10]
    anchors = [[10, 10, 10, 10], [20, 10, 10, 10], [100, 100, 10, 10], [21, 10, 10, 10], [20.5, 10, 10, 10]]
    got = coco.calc_iou(box, anchors, [100] * 5).cpu().numpy()
    np.testing.assert_array_equal(got, [121 / 79, 11 / 189, 0.0, 0.0, 5.5 / 194.5])
    assert coco.calc_iou([10, 20, 4, 8], [[12, 21, 6, 2]], [12]).item() == 15 / 29
    got = calculate_metrics.calc_ious([[0, 0, 10, 10], [5, 5, 15, 15], [11, 0, 20, 10], [10, 0, 20, 10]], [0, 0, 10, 10])
    np.testing.assert_array_equal(got.cpu().numpy(), [121 / 79, 36 / 164, 0.0, 11 / 189])
    # strict threshold, float32 running maximum: the three tie cases of tests/test_legacy_oracle_kat.py
    gt = [[5, 5, 10, 10], [5, 5, 10, 10]]
    for anchor, area, want in (([10, 10, 9, 9], 128, 1), ([10, 10, 9, 9], 150, 1), ([10, 10, 10, 10], 100, 2), ([10, 10, 9, 9], 200, 0)):
        _, tl, s = coco.match_anchors([anchor], [area], gt, [1, 2])
        assert tl.item() == want and s.item() == want


@pytest.mark.parametrize('mode', [0, 1])
def test_iou_tables_bit_exact(mode):
    rng = np.random.default_rng(5 + mode)
    n, k = 4099, 37
    if mode == 0:
        b = cxcyhw_anchors(rng, n)
        a = cxcyhw_anchors(rng, k)
        area = rng.uniform(10, 1e4, n)                      # an input of its own
        got = coco.calc_iou(a, b, area).cpu().numpy()
        want = np.stack([L.calc_iou(a[i], b, area) for i in range(k)])
    else:
        p0 = rng.uniform(0, 400, (n, 2)); b = np.concatenate([p0, p0 + rng.uniform(1, 120, (n, 2))], axis=1)
        t0 = rng.uniform(0, 400, (k, 2)); a = np.concatenate([t0, t0 + rng.uniform(1, 120, (k, 2))], axis=1)
        got = calculate_metrics.calc_ious(b, a).cpu().numpy()
        want = np.stack([L.calc_ious(b, a[i]) for i in range(k)])
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got, want)
    assert (got > 0).mean() > 0.01


@pytest.mark.parametrize('n,g,pad_from', [(1000, 20, 15), (49104, 64, 40), (777, 300, None)])
def test_sequential_matcher_bit_exact(n, g, pad_from):
    rng = np.random.default_rng(n + g)
    bsz = 3
    anchors = cxcyhw_anchors(rng, n)
    areas = anchors[:, 2] * anchors[:, 3]
    gt = gt_xywh(rng, bsz, g, pad_from=pad_from)
    gt[:, 3] = gt[:, 1]                                     # a duplicated row: the tie rule decides
    # GT rows that sit exactly on anchors, so that positives exist at every size
    for b in range(bsz):
        pick = rng.integers(0, n, 4)
        a = anchors[pick]
        gt[b, 5:9] = np.stack([a[:, 0] - a[:, 3] / 2, a[:, 1] - a[:, 2] / 2, a[:, 3], a[:, 2]], axis=1)
    labels = rng.integers(1, 80, (bsz, g)).astype(np.int32)
    tb, tl, s, mi = coco.match_anchors(anchors, areas, gt, labels, return_max_ious=True)
    for b in range(bsz):
        wb, wl, wm = L.sequential_match(anchors, areas, gt[b], labels[b])
        np.testing.assert_array_equal(tb[b].cpu().numpy(), wb)
        np.testing.assert_array_equal(tl[b].cpu().numpy(), wl)
        np.testing.assert_array_equal(mi[b].cpu().numpy(), wm)
        assert s[b].item() == wl.sum() and wl.sum() > 0
    coco.check_true_labels(s)


def test_no_true_labels_is_the_references_error():
    anchors = [[100, 100, 10, 10]]
    _, tl, s = coco.match_anchors(anchors, [100], [[[0, 0, 2, 2], [-1, -1, -1, -1]]], [[5, -1]])
    assert tl.item() == 0 and s.item() == 0
    with pytest.raises(coco.ProcessingError):
        coco.check_true_labels(s)
    # empty GT list: every anchor comes back unchanged
    tb, tl, s = coco.match_anchors(anchors, [100], np.zeros((2, 0, 4)), np.zeros((2, 0), np.int32))
    assert tb.shape == (2, 1, 4) and (tb.cpu().numpy() == anchors).all() and (tl == 0).all() and (s == 0).all()


def test_confusion_matrix_equals_the_restated_reference():
    rng = np.random.default_rng(11)
    ncls = 6
    preds_json, truth = [], {}
    for i in range(12):
        nt, npred = int(rng.integers(0, 7)), int(rng.integers(0, 9))
        t0 = rng.uniform(0, 300, (nt, 2)); twh = rng.uniform(10, 80, (nt, 2))
        true_anns = [dict(bbox=[*t0[j], *twh[j]], category_id=int(rng.integers(0, ncls))) for j in range(nt)]
        anns = []
        for j in range(npred):
            if nt and rng.random() < 0.7:                   # near a true box
                k = int(rng.integers(0, nt))
                x0, y0 = t0[k] + rng.normal(0, 4, 2)
                w, h = twh[k] * rng.uniform(0.8, 1.2, 2)
            else:
                x0, y0 = rng.uniform(0, 300, 2); w, h = rng.uniform(10, 80, 2)
            anns.append(dict(bbox=[x0, y0, x0 + w, y0 + h], objectness=float(rng.uniform(0.1, 0.99)),
                             class_score=float(rng.uniform(0.1, 0.99)), category_id=int(rng.integers(0, ncls))))
        fn = 'img%d.jpg' % i
        preds_json.append(dict(filename=fn, annotations=anns))
        truth[fn] = true_anns
    m = calculate_metrics.Metric(preds_json, list(range(ncls)), ['c%d' % c for c in range(ncls)])
    ref = L.ConfusionMetric(ncls)
    for fn, true_anns in truth.items():
        m.feed_true_values(fn, true_anns)
        anns = m.fn2ann[fn]
        ref.feed_true_values([a['bbox'] for a in true_anns], [a['category_id'] for a in true_anns],
                             np.array([a['bbox'] for a in anns]).reshape(-1, 4), [a['objectness'] for a in anns],
                             [a['class_score'] for a in anns], [a['category_id'] for a in anns])
    assert m.cm_data == ref.cm_data and len(m.cm_data) > 20
    np.testing.assert_array_equal(m.confusion_matrix(), ref.confusion_matrix())
    for c in range(ncls):
        np.testing.assert_array_equal(m.cmetrics[c].dexterity, ref.dexterity[c])
    assert m.per_entry_stats() == ref.per_entry_stats()
    assert any(t != ncls and p != ncls for t, p in m.cm_data)       # real matches happened
