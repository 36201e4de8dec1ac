"""GPU parity of the ground-truth preparation (detection.py:77-95,97-141,
257-262) against the oracle.  Elementwise fp32 with the reference's op order: bit-exact."""
import numpy as np
import pytest
import torch

from object_detection_amd import detection as D
from oracle import odbox_oracle as O

pytestmark = pytest.mark.gpu


def _dataset(seed, B, max_n, sizes=None):
    rng = np.random.default_rng(seed)
    boxes, labels, hw = [], [], []
    for b in range(B):
        H, W = (int(rng.integers(120, 900)), int(rng.integers(120, 900))) if sizes is None else sizes[b]
        n = int(rng.integers(0, max_n + 1))
        x0 = rng.uniform(0, W - 8, n)
        y0 = rng.uniform(0, H - 8, n)
        w = rng.uniform(1, W - x0)
        h = rng.uniform(1, H - y0)
        boxes.append(np.stack([x0, y0, w, h], 1).astype(np.float32).reshape(-1, 4))
        labels.append(rng.integers(0, 80, n).astype(np.int32))
        hw.append((H, W))
    return boxes, labels, hw


def _ref(boxes, labels, hw, image_size, max_items, dtype=np.float32, **kw):
    ex = [O.prepare_example_boxes(bb, ll, s, image_size, max_items, dtype, **kw) for bb, ll, s in zip(boxes, labels, hw)]
    return O.padded_batch_boxes(ex, max_items)


@pytest.mark.parametrize('image_size,max_items,B,max_n', [(512, 64, 9, 40), (416, 20, 6, 60), (1024, 64, 3, 5)])
def test_prepare_batch_equals_oracle(image_size, max_items, B, max_n):
    boxes, labels, hw = _dataset(image_size + B, B, max_n)
    boxes[0] = np.zeros((0, 4), np.float32)                     # an image without annotations
    labels[0] = np.zeros(0, np.int32)
    rb, rl = _ref(boxes, labels, hw, image_size, max_items)
    gb, gl = D.prepare_batch(boxes, labels, hw, image_size, max_items)
    np.testing.assert_array_equal(gb.cpu().numpy(), rb)
    np.testing.assert_array_equal(gl.cpu().numpy(), rl)
    assert (gl[0] == -1).all() and (gb[0] == -1).all()
    if max_n > max_items:
        assert any(len(l) > max_items for l in labels)          # truncation was exercised


def test_half_precision_pipeline_and_output_layouts():
    boxes, labels, hw = _dataset(5, 4, 30)
    rb, rl = _ref(boxes, labels, hw, 512, 64, np.float16)
    gb, gl = D.prepare_batch(boxes, labels, hw, 512, 64, dtype=torch.float16)
    assert gb.dtype == torch.float16
    np.testing.assert_array_equal(gb.float().cpu().numpy(), rb)
    for kw in (dict(want_yx=True), dict(want_wh=True)):
        for np_dt, t_dt in ((np.float32, torch.float32), (np.float16, torch.float16)):
            rb, _ = _ref(boxes, labels, hw, 512, 64, np_dt, **kw)
            gb, _ = D.prepare_batch(boxes, labels, hw, 512, 64, dtype=t_dt, **kw)
            valid = rl != -1
            np.testing.assert_array_equal(gb.float().cpu().numpy()[valid], rb[valid])


def test_known_answer_square_pad_and_scale():
    # 300 x 400 image -> 400 canvas, y offset 50; box (10, 20, 30, 40) -> corners x 10..40, y 70..110;
    # / 400 * 512 -> (12.8, 89.6, 51.2, 140.8)
    gb, gl = D.prepare_batch([[[10, 20, 30, 40]]], [[7]], [(300, 400)], 512, 4)
    np.testing.assert_allclose(gb[0, 0].cpu().numpy(), [12.8, 89.6, 51.2, 140.8], rtol=1e-6)
    assert gl.cpu().tolist() == [[7, -1, -1, -1]] and (gb[0, 1:] == -1).all()


def test_polygons_after_augmentation_and_polygon2bbox():
    rng = np.random.default_rng(3)
    polys = [rng.uniform(0, 600, (n, 4, 2)).astype(np.float32) for n in (5, 0, 70)]
    labels = [rng.integers(0, 9, len(p)).astype(np.int32) for p in polys]
    canvas = [600.0, 512.0, 640.0]
    ex = [O.prepare_polygons(p, l, c, 512, 64) for p, l, c in zip(polys, labels, canvas)]
    rb, rl = O.padded_batch_boxes(ex, 64)
    gb, gl = D.prepare_batch_polygons(polys, labels, canvas, 512, 64)
    np.testing.assert_array_equal(gb.cpu().numpy(), rb)
    np.testing.assert_array_equal(gl.cpu().numpy(), rl)
    for kw in (dict(), dict(want_yx=True), dict(want_wh=True)):
        np.testing.assert_array_equal(D.polygon2bbox(polys[2], **kw).cpu().numpy(), O.polygon2bbox(polys[2], **kw))
    h = torch.as_tensor(polys[2]).half()
    np.testing.assert_array_equal(D.polygon2bbox(h, want_wh=True).float().cpu().numpy(),
                                  O.polygon2bbox(h.numpy(), want_wh=True).astype(np.float32))


def test_prepared_batch_feeds_the_matcher():
    from object_detection_amd import anchors as A
    boxes, labels, hw = _dataset(9, 4, 20)
    gb, gl = D.prepare_batch(boxes, labels, hw, 256, 64)
    anc = O.generate_anchors(256)
    reg, lab = A.anchor_targets_bbox(anc, torch.zeros((4, 256, 256, 3)), gb, gl, 80)
    rb, rl = _ref(boxes, labels, hw, 256, 64)
    ref = O.anchor_targets_bbox(anc, (256, 256), rb, rl, 80)
    np.testing.assert_array_equal(lab.cpu().numpy(), ref['labels'])
    np.testing.assert_allclose(reg.cpu().numpy(), ref['regression'], rtol=1e-5, atol=1e-6)


def test_bad_arguments():
    with pytest.raises(ValueError):
        D.prepare_batch([np.zeros((2, 4))], [np.zeros(3)], [(10, 10)], 512)
    with pytest.raises(ValueError):
        D.prepare_batch([np.zeros((2, 4))], [np.zeros(2)], [(10, 10)], 512, dtype=torch.float64)


def test_corrupted_row_offsets_give_wrong_rows_never_a_fault():
    """The offsets table is read on the device: values outside [0, rows] are clamped there (include/odbox.h)."""
    from object_detection_amd import _lib
    lib = _lib.load()
    dev = _lib.require_gpu()
    rows, B, M = 10, 4, 8
    xywh = torch.rand((rows, 4), device=dev) * 50 + 1
    lab = torch.arange(rows, dtype=torch.int32, device=dev)
    hw = torch.full((B, 2), 100.0, device=dev)
    ob = torch.empty((B, M, 4), device=dev)
    ol = torch.empty((B, M), dtype=torch.int32, device=dev)
    offs = torch.tensor([-5, 3, 1 << 30, -(1 << 30), 7], dtype=torch.int32, device=dev)
    rc = lib.odbox_gt_prepare_xywh(xywh.data_ptr(), lab.data_ptr(), offs.data_ptr(), rows, hw.data_ptr(), B, M, 64.0, 0, 0,
                                   ob.data_ptr(), ol.data_ptr(), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    got = ol.cpu().numpy()
    np.testing.assert_array_equal(got[0], [0, 1, 2, -1, -1, -1, -1, -1])       # [-5, 3) -> [0, 3)
    np.testing.assert_array_equal(got[1], [3, 4, 5, 6, 7, 8, 9, -1])           # [3, 2^30) -> [3, 10)
    assert (got[2] == -1).all()                                                # [2^30, -2^30) -> [10, 10): empty
    np.testing.assert_array_equal(got[3], [0, 1, 2, 3, 4, 5, 6, -1])           # [-2^30, 7) -> [0, 7)
