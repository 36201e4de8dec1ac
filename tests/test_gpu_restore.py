"""GPU parity of the output side of inference (infdet.py:140-174: detections of the padded,
resized square back to original image pixels) against the oracle: fp64 arithmetic, bit-exact,
and the per-image count where the reference's loop breaks (first score <= 0)."""
import numpy as np
import pytest
import torch

from object_detection_amd import bndbox as BB
from object_detection_amd import detection as D
from object_detection_amd import workloads
from oracle import odbox_oracle as O

pytestmark = pytest.mark.gpu


def _case(seed, B, R, image_size):
    rng = np.random.default_rng(seed)
    coords = rng.uniform(0, image_size, (B, R, 4)).astype(np.float32)
    scores = np.sort(rng.uniform(0.01, 1, (B, R)).astype(np.float32), axis=1)[:, ::-1].copy()
    counts = rng.integers(0, R + 1, B)
    counts[0], counts[-1] = 0, R                         # an empty image and a full one
    for b in range(B):
        scores[b, counts[b]:] = 0
    hw = np.stack([rng.integers(17, 4000, B), rng.integers(17, 4000, B)], 1).astype(np.int32)
    hw[0] = (640, 640)                                   # no padding at all
    return coords, scores, hw, counts


@pytest.mark.parametrize('B,R,image_size', [(5, 100, 512), (1, 1, 416), (3, 700, 1024), (64, 100, 640)])
def test_restore_matches_oracle_bit_for_bit(B, R, image_size):
    coords, scores, hw, counts = _case(B + R, B, R, image_size)
    out, cnt = D.restore_detections(coords, scores, hw, image_size)
    ref, rcnt = O.restore_detections(coords, scores, hw, image_size)
    assert out.dtype == torch.float64
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    np.testing.assert_array_equal(rcnt, counts)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), ref.view(np.uint64))


def test_break_at_first_nonpositive_score_even_if_later_ones_are_positive():
    coords, scores, hw, _ = _case(3, 4, 50, 512)
    scores[:] = 0.5
    scores[1, 7] = 0.0                                   # the loop breaks here; later rows are never looked at
    scores[2, 0] = -1.0
    scores[3, 49] = 0.0
    out, cnt = D.restore_detections(coords, scores, hw, 512)
    ref, rcnt = O.restore_detections(coords, scores, hw, 512)
    np.testing.assert_array_equal(cnt.cpu().numpy(), [50, 7, 0, 49])
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), ref.view(np.uint64))


def test_end_to_end_decode_nms_restore_json():
    """decode + NMS -> restore -> the results.json structure of infdet.py:152-188."""
    from object_detection_amd import anchors as A
    S, B, C = 256, 3, 6
    anchors = A.generate_anchors(A.AnchorsConfig(), S)
    reg, sc = workloads.nms_inputs(anchors.shape[0], C=C, batch=B, seed=5)
    boxes, scores, labels = BB.decode_and_nms(anchors, reg, sc, (S, S))
    hw = np.array([(480, 640), (1080, 1920), (300, 200)], np.int32)
    out, cnt = D.restore_detections(boxes, scores, hw, S)
    ref, rcnt = O.restore_detections(boxes.cpu().numpy(), scores.cpu().numpy(), hw, S)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), ref.view(np.uint64))
    js = D.results_json(['a.jpg', 'b.jpg', 'c.jpg'], out, scores, labels, cnt)
    assert [len(j['annotations']) for j in js] == list(rcnt)
    assert sum(rcnt) > 0
    b = int(np.argmax(rcnt))
    a0 = js[b]['annotations'][0]
    assert a0['bbox'] == [float(v) for v in ref[b, 0]]
    assert a0['class_score'] == float(scores[b, 0]) and a0['category_id'] == int(labels[b, 0])
    assert js[0]['filename'] == 'a.jpg'


def test_empty_batch_and_bad_arguments():
    out, cnt = D.restore_detections(np.zeros((0, 10, 4), np.float32), np.zeros((0, 10), np.float32), np.zeros((0, 2), np.int32), 512)
    assert out.shape == (0, 10, 4) and cnt.shape == (0,)
    with pytest.raises(ValueError):
        D.restore_detections(np.zeros((2, 10, 4), np.float32), np.zeros((2, 9), np.float32), np.zeros((2, 2), np.int32), 512)
