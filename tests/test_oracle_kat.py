"""Known-answer tests of the CPU oracle (hand-derived) and cross-check of its two
independent restatements (NumPy, C).  The reference holds no fixtures for this path, so these
are what pins the oracle ("parity unpinned" against the reference itself)."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import odbox_oracle as O

F = np.float32


def test_base_anchor_table_size32():
    a = O.base_anchors(32, (1, 2, .5))
    assert a.shape == (9, 4) and a.dtype == np.float32
    np.testing.assert_array_equal(a[0], F([-16, -16, 16, 16]))
    np.testing.assert_allclose(a[1], [-20.158737, -20.158737, 20.158737, 20.158737], rtol=0, atol=2e-6)
    np.testing.assert_allclose(a[2], [-25.398417, -25.398417, 25.398417, 25.398417], rtol=0, atol=2e-6)
    np.testing.assert_allclose(a[3], [-11.313708, -22.627417, 11.313708, 22.627417], rtol=0, atol=2e-6)
    np.testing.assert_allclose(a[8], [-35.918785, -17.959393, 35.918785, 17.959393], rtol=0, atol=3e-6)


@pytest.mark.parametrize('S,N', [(512, 49104), (1024, 196416)])
def test_generate_anchors_counts_and_corners(S, N):
    a = O.generate_anchors(S)
    assert a.shape == (N, 4)
    np.testing.assert_array_equal(a[0], F([-12, -12, 20, 20]))
    if S == 512:
        np.testing.assert_allclose(a[-1], [-126.70056, 160.64972, 1022.70056, 735.3503], rtol=1e-6)
    xc = (a[:, 0] + a[:, 2]) / 2
    yc = (a[:, 1] + a[:, 3]) / 2
    assert (xc < S).all() and (yc < S).all()


def test_yolo_shaped_grid():
    assert O.yolo_shaped_anchors(416).shape == (10647, 4)


def test_tile_order_y_outer_x_inner():
    base = O.base_anchors(32, [1])
    t = O.tile_anchors(base, 2, 3, 8)
    assert t.shape == (2 * 3 * 3, 4)
    # cell (i=0, j=1) -> rows 3..5, shifted by (12, 4)
    np.testing.assert_array_equal(t[3], base[0] + F([12, 4, 12, 4]))
    # cell (i=1, j=0) -> rows 9..11, shifted by (4, 12)
    np.testing.assert_array_equal(t[9], base[0] + F([4, 12, 4, 12]))
    np.testing.assert_array_equal(t, c_oracle.anchor_tile(base, 2, 3, 8))


def _iou(a, g):
    return O.bbox_overlap(F(a).reshape(1, 1, 4), F(g).reshape(1, 1, 4))[0, 0, 0]


def test_iou_known_answers():
    assert _iou([0, 0, 10, 10], [0, 0, 10, 10]) == F(1.0)
    assert _iou([0, 0, 10, 10], [20, 20, 30, 30]) == F(0.0)
    assert _iou([0, 0, 10, 10], [5, 0, 15, 10]) == F(50.0) / F(150.0)
    assert _iou([0, 0, 10, 10], [5, 0, 15, 10]) == F(0.33333334)
    assert _iou([0, 0, 10, 10], [-1, -1, -1, -1]) == F(-1.0)


def _targets(anchors, gt, labels, C=3, hw=(512, 512)):
    return O.anchor_targets_bbox(F(anchors), hw, F(gt)[None], np.asarray(labels, np.int32)[None], C)


def test_threshold_edges():
    anchors = [[0, 0, 10, 10]]
    # IoU exactly 0.5 -> positive (>=)
    r = _targets(anchors, [[0, 0, 10, 5]], [2])
    assert r['state'][0, 0] == 1 and r['labels'][0, 0].tolist() == [0, 0, 1, 1]
    # IoU 0.4f is not > 0.4f -> background
    r = _targets(anchors, [[0, 0, 10, 4]], [2])
    assert r['max_iou'][0, 0] == F(0.4) and r['state'][0, 0] == 0
    assert r['labels'][0, 0].tolist() == [0, 0, 0, 0]
    # IoU in (0.4, 0.5) -> ignore
    r = _targets(anchors, [[0, 0, 10, 4.5]], [2])
    assert r['state'][0, 0] == -1 and r['labels'][0, 0].tolist() == [0, 0, 0, -1]


def test_ties_and_padding():
    anchors = [[0, 0, 10, 10], [100, 100, 110, 110]]
    gt = [[0, 0, 10, 10], [0, 0, 10, 10], [-1, -1, -1, -1]]
    r = _targets(anchors, gt, [1, 2, -1])
    assert r['match_idx'][0].tolist() == [0, 0]          # identical rows: lower index; no overlap: 0
    assert r['state'][0].tolist() == [1, 0]
    assert r['labels'][0, 0].tolist() == [0, 1, 0, 1]
    # all rows padded: argmax 0, max -1, every anchor ignored, one-hot all zero
    r = _targets(anchors, [[-1] * 4] * 3, [-1] * 3)
    assert r['match_idx'][0].tolist() == [0, 0]
    assert (r['max_iou'][0] == -1).all() and (r['state'][0] == -1).all()
    assert (r['labels'][0, :, :3] == 0).all() and (r['labels'][0, :, 3] == -1).all()


def test_out_of_image_anchor_is_ignored():
    anchors = [[500, 10, 530, 40]]           # centre x = 515 >= 512
    r = _targets(anchors, [[500, 10, 530, 40]], [0])
    assert r['max_iou'][0, 0] == F(1.0) and r['state'][0, 0] == -1
    r = _targets(anchors, [[500, 10, 530, 40]], [0], hw=(512, 600))
    assert r['state'][0, 0] == 1


def test_encode_known_answer_and_decode_inverse():
    a = F([[0, 0, 10, 20]])
    g = F([[[1, 2, 11, 18]]])
    t = O.bbox_transform(a[None], g)
    np.testing.assert_allclose(t[0, 0], [0.5, 0.5, 0.5, -0.5], rtol=1e-6)
    back = O.regress_bndboxes(a[None], t)
    np.testing.assert_allclose(back[0, 0], g[0, 0], rtol=1e-6)


def test_returns_regression_for_every_anchor():
    # negatives / ignored / padded matches still get offsets (anchors.py:199-202)
    r = _targets([[0, 0, 10, 10]], [[-1, -1, -1, -1]], [-1])
    np.testing.assert_allclose(r['regression'][0, 0, :4], [-0.5, -0.5, -5.5, -5.5], rtol=1e-6)
    assert r['regression'][0, 0, 4] == -1


def test_clip():
    b = F([[[-5, -1, 600, 511.5]]])
    np.testing.assert_array_equal(O.clip_boxes(b, (512, 512))[0, 0], F([0, 0, 511, 511]))


def _nms1(boxes, scores, **kw):
    return O.nms(F(boxes)[None], F(scores)[None], **kw)


def test_nms_known_answers():
    box = [0, 0, 10, 10]
    # duplicates, same class -> one kept; lower anchor index first on equal score
    b, s, l, k, n = _nms1([box, box], [[0.9, 0.1], [0.9, 0.1]])
    assert n[0] == 1 and k[0, 0] == 0 and l[0, 0] == 0 and s[0, 0] == F(0.9)
    assert (s[0, 1:] == 0).all() and (l[0, 1:] == 0).all() and (b[0, 1:] == 0).all() and (k[0, 1:] == -1).all()
    # duplicates in different classes -> both kept, top_k tie -> lower class first
    b, s, l, k, n = _nms1([box, box], [[0.9, 0.1], [0.1, 0.9]])
    assert n[0] == 2 and l[0, :2].tolist() == [0, 1] and k[0, :2].tolist() == [0, 1]
    # score == threshold -> dropped
    b, s, l, k, n = _nms1([box], [[0.3, 0.1]])
    assert n[0] == 0 and (s == 0).all()
    # IoU exactly == threshold -> kept (suppress only when strictly greater): 50/100 = 0.5
    b, s, l, k, n = _nms1([[0, 0, 10, 10], [0, 0, 10, 5]], [[0.9], [0.8]], iou_threshold=0.5)
    assert n[0] == 2
    b, s, l, k, n = _nms1([[0, 0, 10, 10], [0, 0, 10, 5]], [[0.9], [0.8]], iou_threshold=0.49)
    assert n[0] == 1
    # zero-area box never suppresses nor is suppressed
    b, s, l, k, n = _nms1([[5, 5, 5, 9], [5, 5, 5, 9]], [[0.9], [0.8]])
    assert n[0] == 2
    # no detections -> all zero
    b, s, l, k, n = _nms1([box], [[0.1, 0.2]])
    assert n[0] == 0 and (b == 0).all() and (s == 0).all() and (l == 0).all()


def test_nms_per_class_truncation_then_global_topk():
    # 6 disjoint boxes in class 0 and 2 in class 1; max_ret 3 -> per class cap 3, then top 3 overall
    boxes = [[20 * i, 0, 20 * i + 10, 10] for i in range(8)]
    scores = [[0.50 + 0.01 * i, 0] for i in range(6)] + [[0, 0.95], [0, 0.40]]
    b, s, l, k, n = _nms1(boxes, scores, max_ret=3)
    assert n[0] == 3
    assert k[0].tolist() == [6, 5, 4] and l[0].tolist() == [1, 0, 0]


def _rand_match_case(rng, B, N, G, C, S=256):
    cx, cy = rng.uniform(0, S, (2, N))
    w, h = rng.uniform(4, S / 2, (2, N))
    anchors = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(F)
    gx, gy = rng.uniform(0, S, (2, B, G))
    gw, gh = rng.uniform(4, S / 2, (2, B, G))
    gt = np.stack([gx - gw / 2, gy - gh / 2, gx + gw / 2, gy + gh / 2], -1).astype(F)
    # make some GT identical to anchors (IoU 1, ties) and some rows padded
    gt[:, 0] = anchors[:B]
    gt[:, 1] = anchors[:B]
    labels = rng.integers(0, C, (B, G)).astype(np.int32)
    npad = rng.integers(0, G // 2, B)
    for b in range(B):
        if npad[b]:
            gt[b, G - npad[b]:] = -1
            labels[b, G - npad[b]:] = -1
    return anchors, gt, labels


def test_c_and_numpy_match_encode_agree():
    rng = np.random.default_rng(7)
    anchors, gt, labels = _rand_match_case(rng, 3, 700, 12, 5)
    ref = O.anchor_targets_bbox(anchors, (256, 200), gt, labels, 5)
    got = c_oracle.match_encode(anchors, (256, 200), gt, labels, 5, threads=2)
    for key in ('match_idx', 'state', 'max_iou', 'labels', 'regression'):
        np.testing.assert_array_equal(ref[key], got[key], err_msg=key)
    np.testing.assert_array_equal(O.bbox_overlap(np.broadcast_to(anchors, (3,) + anchors.shape), gt),
                                  c_oracle.bbox_overlap(anchors, gt))


def test_c_and_numpy_decode_nms_agree():
    rng = np.random.default_rng(11)
    N, C, B = 900, 7, 2
    anchors = O.tile_anchors(O.base_anchors(32, (1, 2, .5)), 10, 10, 8)
    regress = np.clip(rng.normal(0, 0.5, (B, N, 4)), -2, 2).astype(F)
    scores = rng.uniform(0, 0.25, (B, N, C)).astype(F)
    hot = rng.integers(0, N, (B, 200))
    for b in range(B):
        scores[b, hot[b], rng.integers(0, C, 200)] = (np.floor((0.3 + 0.7 * rng.uniform(size=200)) * 64) / 64).astype(F)
    ref = O.decode_and_nms(anchors, regress, scores, (80, 80), max_ret=20)
    got = c_oracle.decode_nms(anchors, regress, scores, (80, 80), max_ret=20, threads=2)
    for r, g, name in zip(ref, got, ('boxes', 'scores', 'labels', 'keep', 'count')):
        np.testing.assert_array_equal(r, g, err_msg=name)
    assert ref[4].min() > 0


def test_focal_and_huber_known_answers():
    # y=1, p=0.5: -0.25 * 0.5^1.5 * ln 0.5 ; y=0, p=0.5: -0.75 * 0.5^1.5 * ln 0.5
    v = O.focal_loss_elements(F([1, 0]), F([0.5, 0.5]))
    np.testing.assert_allclose(v, [0.25 * 0.5 ** 1.5 * np.log(2), 0.75 * 0.5 ** 1.5 * np.log(2)], rtol=1e-6)
    # clipping: p = 0 with y = 1 -> p = 1e-6
    v = O.focal_loss_elements(F([1]), F([0.0]))
    np.testing.assert_allclose(v, [-0.25 * (1 - 1e-6) ** 1.5 * np.log(1e-6)], rtol=1e-5)
    # Huber(delta=1): errors (0.5, -2, 0, 1) -> (0.125, 1.5, 0, 0.5) -> mean 0.53125
    np.testing.assert_allclose(O.huber(F([[0, 0, 0, 0]]), F([[0.5, -2, 0, 1]])), [0.53125], rtol=1e-7)


def test_model_metric_normaliser_and_gradient_consistency():
    rng = np.random.default_rng(21)
    anchors = O.generate_anchors(128)
    N = anchors.shape[0]
    gt = np.full((1, 4, 4), -1, F)
    gt[0, 0] = anchors[300]
    lab = np.int32([[2, -1, -1, -1]])
    pb = rng.normal(0, 1, (1, N, 4)).astype(F)
    ps = rng.uniform(0.05, 0.95, (1, N, 3)).astype(F)
    d, c, n, gb, gs = O.model_metric(anchors, (128, 128), gt, lab, pb, ps, 3, with_grads=True)
    assert n >= 1
    # finite-difference check of the analytic class gradient on one element
    t = O.anchor_targets_bbox(anchors, (128, 128), gt, lab, 3)
    i = int(np.nonzero(t['state'][0] == 1)[0][0])
    for k in range(3):
        h = 1e-3
        up, dn = ps.copy(), ps.copy()
        up[0, i, k] += h
        dn[0, i, k] -= h
        num = (O.model_metric(anchors, (128, 128), gt, lab, pb, up, 3)[1] -
               O.model_metric(anchors, (128, 128), gt, lab, pb, dn, 3)[1]) * n / (2 * h)
        np.testing.assert_allclose(gs[0, i, k], num, rtol=2e-3)


def test_gt_preparation_known_answers():
    """detection.py:97-141: 300x400 image -> 400 canvas (y offset 50), rescale to 512."""
    b, l = O.prepare_example_boxes([[10, 20, 30, 40], [0, 0, 400, 300]], [7, 3], (300, 400), 512, 64)
    np.testing.assert_allclose(b, [[12.8, 89.6, 51.2, 140.8], [0, 64, 512, 448]], rtol=1e-6)
    assert l.tolist() == [7, 3] and b.dtype == np.float32
    b, l = O.prepare_example_boxes(np.tile([[1, 1, 2, 2]], (5, 1)), np.arange(5), (100, 100), 100, 3)
    assert b.shape == (3, 4) and l.tolist() == [0, 1, 2]                # cut at max_items_in_image
    pb, pl = O.padded_batch_boxes([(b, l), (np.zeros((0, 4)), np.zeros(0))], 4)
    assert pb.shape == (2, 4, 4) and (pb[0, 3] == -1).all() and (pb[1] == -1).all() and pl[1].tolist() == [-1] * 4
    poly = np.array([[[3, 9], [7, 1], [5, 5], [2, 4]]], np.float32)     # a rotated quadrilateral
    assert O.polygon2bbox(poly).tolist() == [[2, 1, 7, 9]]
    assert O.polygon2bbox(poly, want_yx=True).tolist() == [[1, 2, 9, 7]]
    assert O.polygon2bbox(poly, want_wh=True).tolist() == [[2, 1, 5, 8]]
    h, _ = O.prepare_example_boxes([[10, 20, 30, 40]], [1], (300, 400), 512, 4, np.float16)
    assert h.dtype == np.float16 and h.tolist() == [[12.796875, 89.625, 51.1875, 140.75]]


def test_restore_detections_known_answers():
    """infdet.py:140-174 by hand: a 480x640 image seen as a 512 square has max_side 640, pad_y 80,
    pad_x 0, scale 1.25; a 300x200 one has max_side 300, pad_x 50, scale 300/512."""
    from oracle import odbox_oracle as O
    coords = np.array([[[64, 128, 256, 384], [0, 0, 512, 512]], [[100, 100, 200, 200], [1, 2, 3, 4]]], np.float32)
    scores = np.array([[0.9, 0.5], [0.7, 0.0]], np.float32)
    out, cnt = O.restore_detections(coords, scores, [(480, 640), (300, 200)], 512)
    assert list(cnt) == [2, 1]
    np.testing.assert_array_equal(out[0], [[80.0, 80.0, 320.0, 400.0], [0.0, -80.0, 640.0, 560.0]])
    s = 300 / 512
    np.testing.assert_array_equal(out[1, 0], [100 * s - 50, 100 * s, 200 * s - 50, 200 * s])
    np.testing.assert_array_equal(out[1, 1], [0, 0, 0, 0])


def test_gt_best_anchor_known_answers_and_two_restatements_agree():
    """Per-GT best anchor (the extra output of odbox_match_encode): hand-derived cases, NumPy vs C."""
    from oracle import c_oracle
    anchors = np.array([[0, 0, 10, 10], [0, 0, 10, 5], [0, 0, 10, 10], [20, 20, 30, 30]], np.float32)
    gt = np.array([[[0, 0, 10, 10],        # IoU 1 with anchors 0 and 2: the lower one wins
                    [0, 0, 10, 4],         # 40/50 with anchor 1 (0.8) beats 40/100
                    [100, 100, 110, 110],  # touches nothing: all +0 -> anchor 0
                    [-1, -1, -1, -1]]], np.float32)   # padding: all -1 -> anchor 0
    idx, iou = O.gt_best_anchors(anchors, gt)
    np.testing.assert_array_equal(idx, [[0, 1, 0, 0]])
    np.testing.assert_allclose(iou, [[1.0, 0.8, 0.0, -1.0]], rtol=1e-6)
    cidx, ciou = c_oracle.gt_best(anchors, gt)
    np.testing.assert_array_equal(cidx, idx)
    np.testing.assert_array_equal(ciou, iou)
    from object_detection_amd import workloads
    rng = np.random.default_rng(4)
    a = O.generate_anchors(128)
    g = np.stack([workloads.gt_boxes(rng, 128, 9, 3, pad_to=12)[0] for _ in range(3)])
    i1, v1 = O.gt_best_anchors(a, g)
    i2, v2 = c_oracle.gt_best(a, g, threads=2)
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(v1, v2)


def tf_nms_cases():
    import json
    import os
    doc = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'tf_nms_vectors.json')))
    return doc['cases']


def test_tf_nms_restatement_against_tensorflows_own_unit_test_vectors():
    """Row a11: TensorFlow's NonMaxSuppressionV3 is a third-party kernel that is absent here; its restatement
    (oracle tf_non_max_suppression) is pinned by the known answers of TensorFlow's own unit tests
    (tests/golden/tf_nms_vectors.json names the source)."""
    for name, (boxes, scores, max_out, iou_thr, score_thr, want) in tf_nms_cases().items():
        got = O.tf_non_max_suppression(np.float32(boxes), np.float32(scores), max_out, iou_thr, score_thr)
        assert got.tolist() == want, name
        # the same vector through the whole oracle nms() (one class, xyxy columns) and its C twin
        xyxy = np.float32(boxes)[None][..., [1, 0, 3, 2]]
        sc = np.float32(scores)[None, :, None]
        _, _, _, keep, count = O.nms(xyxy, sc, score_thr, iou_thr, max_out)
        assert keep[0, :count[0]].tolist() == want, name
        _, _, _, keep, count = c_oracle.decode_nms(None, xyxy, sc, (200, 200), do_clip=False, score_threshold=score_thr,
                                                   iou_threshold=iou_thr, max_ret=max_out)
        assert keep[0, :count[0]].tolist() == want, name


def test_huber_restatement_against_the_keras_docstring_values():
    """tf.keras.losses.Huber is third party (call site metric.py:99); the restatement is held to the values its
    published docstring prints (tests/golden/keras_huber_vectors.json names the source)."""
    import json
    import os
    v = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'keras_huber_vectors.json')))
    per_sample = O.huber(F(v['y_true']), F(v['y_pred']), v['delta'])
    np.testing.assert_allclose(per_sample, v['reduction_none'], atol=5e-4)           # exact: 0.5*(0.36+0.36)/2, 0.5*(0.16+0.36)/2
    np.testing.assert_allclose(per_sample, [0.18, 0.13], rtol=1e-6)
    assert abs(per_sample.sum() - v['reduction_sum']) < 5e-4
    assert abs(per_sample.mean() - v['reduction_sum_over_batch_size']) < 5e-4
    assert abs((per_sample * [1, 0]).mean() - v['sample_weight_1_0']) < 5e-4
