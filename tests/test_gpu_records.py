"""The detection records of the multi-GPU all-gather: the HIP pack / unpack used for
GPU tensors must produce exactly what the torch path used on CPU tensors (the gloo tests) produces,
bit for bit, labels included (they travel bit-cast inside a float tensor)."""
import numpy as np
import pytest
import torch

from object_detection_amd import dist as D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('b,R,rows', [(3, 100, None), (1, 1, 4), (5, 37, 8), (64, 100, 64)])
def test_pack_unpack_match_the_cpu_path_bit_for_bit(b, R, rows):
    rng = np.random.default_rng(b * 131 + R)
    boxes = torch.as_tensor(rng.normal(0, 300, (b, R, 4)).astype(np.float32))
    scores = torch.as_tensor(rng.random((b, R)).astype(np.float32))
    labels = torch.as_tensor(rng.integers(-3, 1 << 30, (b, R)).astype(np.int32))
    labels[0, 0] = 0x7FC00001                      # would be a NaN payload as a float: bits must survive
    want = D.pack_detections(boxes, scores, labels, rows=rows)
    got = D.pack_detections(boxes.cuda(), scores.cuda(), labels.cuda(), rows=rows)
    assert got.is_cuda and got.shape == want.shape
    assert torch.equal(got.cpu().view(torch.int32), want.view(torch.int32))
    ub, us, ul = D.unpack_detections(got)
    n = want.shape[0]
    assert torch.equal(ub[:b].cpu(), boxes) and torch.equal(us[:b].cpu(), scores) and torch.equal(ul[:b].cpu(), labels)
    assert (ub[b:n] == 0).all() and (ul[b:n] == 0).all()
    wb, ws, wl = D.unpack_detections(want)
    assert torch.equal(wb, ub.cpu()) and torch.equal(ws, us.cpu()) and torch.equal(wl, ul.cpu())


def test_single_process_gather_is_the_identity():
    boxes, scores, labels = torch.zeros((2, 5, 4)).cuda(), torch.ones((2, 5)).cuda(), torch.ones((2, 5), dtype=torch.int32).cuda()
    out = D.gather_detections(boxes, scores, labels, batch=2)
    assert out[0] is boxes and out[1] is scores and out[2] is labels
