"""match_encode_sweep_kernel (streamed dense labels written as a chip-wide lockstep sweep by dedicated store
waves, matcher waves patching behind them): the same bits as the tile kernels and as the CPU oracle, through
the C-ABI.  `odbox_match_encode_plan` says which kernel a shape takes, so every case here asserts that it
really ran the sweep."""
import numpy as np
import pytest
import torch

from object_detection_amd import _lib
from object_detection_amd import anchors as A
from object_detection_amd import workloads
from oracle import c_oracle
from oracle import odbox_oracle as O

pytestmark = pytest.mark.gpu
F = np.float32


def cpu(t):
    return t.detach().cpu().numpy()


@pytest.fixture
def stream_policy():
    prev = A.set_output_policy('stream')
    yield
    A.set_output_policy(prev)


def both_policies(anchors, hw, gt, lab, C, **kw):
    """All outputs under 'stream' (the sweep) and under 'cache' (256-anchor tiles)."""
    B, G = gt.shape[:2]
    assert A.match_encode_plan(B, anchors.shape[0], G, C) == 'sweep'
    sweep = A.match_encode(anchors, hw, gt, lab, C, dense=True, compact=True, **kw)
    prev = A.set_output_policy('cache')
    try:
        assert A.match_encode_plan(B, anchors.shape[0], G, C) == 'tiles'
        tiles = A.match_encode(anchors, hw, gt, lab, C, dense=True, compact=True, **kw)
    finally:
        A.set_output_policy(prev)
    return sweep, tiles


def check(anchors, hw, gt, lab, C, oracle=True):
    sweep, tiles = both_policies(anchors, hw, gt, lab, C)
    for k in tiles:
        a, b = sweep[k], tiles[k]
        # the same values everywhere (NaN offsets of zero-size anchors included); the one freedom is the sign of a
        # zero maximum over rows that mix +0 and -0 (negative-area anchors), which the oracle comparison below
        # does not fix either
        same = (a == b) | (a.isnan() & b.isnan()) if a.dtype == torch.float32 else a == b
        if not bool(same.all()):
            bad = (~same).nonzero()
            where = tuple(bad[0].tolist())
            raise AssertionError('%s: %d elements differ, first at %s: %r vs %r' % (
                k, bad.shape[0], where, sweep[k][where].item(), tiles[k][where].item()))
    if oracle:
        ref = c_oracle.match_encode(anchors, hw, gt, lab, C, 0.4, 0.5, threads=8)
        np.testing.assert_array_equal(cpu(sweep['match_idx']), ref['match_idx'])
        np.testing.assert_array_equal(cpu(sweep['state']), ref['state'])
        np.testing.assert_array_equal(cpu(sweep['max_iou']), ref['max_iou'])
        np.testing.assert_array_equal(cpu(sweep['labels']), ref['labels'])
        reg = cpu(sweep['regression'])
        np.testing.assert_array_equal(reg[..., 4], ref['regression'][..., 4])
        with np.errstate(invalid='ignore'):
            np.testing.assert_allclose(reg, ref['regression'], rtol=1e-5, atol=0, equal_nan=True)
    return sweep


def random_case(rng, B, N, G, C, S=256, pads=2):
    cx, cy = rng.uniform(0, S, (2, N))
    w, h = rng.uniform(4, S / 2, (2, N))
    anchors = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(F)
    gx, gy = rng.uniform(0, S, (2, B, G))
    gw, gh = rng.uniform(4, S / 2, (2, B, G))
    gt = np.stack([gx - gw / 2, gy - gh / 2, gx + gw / 2, gy + gh / 2], -1).astype(F)
    k = min(N, G, 4)
    gt[:, :k] = anchors[:k]                      # exact matches
    if G > 2:
        gt[:, 2] = gt[:, 0]                      # duplicate row -> argmax tie: the first wins
    lab = rng.integers(0, C, (B, G)).astype(np.int32)
    if G > 4 and pads:
        gt[:, -pads:] = -1
        lab[:, -pads:] = -1
    return anchors, gt, lab


# rows of C + 1 = 69 .. 201 floats; anchor counts that put row and image boundaries anywhere inside the 4 KB
# pieces; totals that are not a multiple of 4 floats (scalar tail); fewer pieces than blocks; G = 1 .. 64
@pytest.mark.parametrize('B,N,G,C', [(1, 1, 1, 68), (1, 3, 2, 80), (2, 255, 3, 80), (3, 257, 7, 90), (2, 1000, 20, 80),
                                     (1, 513, 64, 69), (5, 4099, 33, 100), (2, 7001, 20, 200), (7, 1237, 5, 71),
                                     (8192, 37, 2, 70)])
def test_random_shapes_bit_exact(stream_policy, B, N, G, C):
    rng = np.random.default_rng(B * 1000 + N + G + C)
    anchors, gt, lab = random_case(rng, B, N, G, C)
    check(anchors, (256, 216), gt, lab, C)


def test_what_takes_the_sweep(stream_policy):
    assert A.match_encode_plan(64, 10647, 20, 80) == 'sweep'
    assert A.match_encode_plan(64, 10647, 65, 80) == 'tiles_stream'            # long GT lists: LDS-tiled matcher
    assert A.match_encode_plan(64, 10647, 20, 67) == 'tiles_stream'            # narrow rows: more than 16 per piece
    assert A.match_encode_plan(64, 10647, 20, 80, labels_aligned16=False) == 'tiles_stream'
    assert A.match_encode_plan(64, 10647, 20, 80, gt_best=True) == 'tiles_stream'
    assert A.match_encode_plan(32, 196416, 20, 80) == 'tiles_stream'           # 2 GB of labels: tiles win there
    A.set_output_policy('auto')
    assert A.match_encode_plan(64, 10647, 20, 80) == 'sweep'                   # AUTO: a buffer that no recent call wrote
    assert A.match_encode_plan(64, 10647, 65, 80) == 'tiles'
    assert A.match_encode_plan(160, 10647, 20, 80) == 'sweep'                  # beyond 512 MB AUTO always streams
    A.set_output_policy('cache')
    assert A.match_encode_plan(64, 10647, 20, 80) == 'tiles'
    with pytest.raises(_lib.OdboxError):
        A.match_encode_plan(1, 1, 0, 80)


def test_padded_rows_and_out_of_range_labels(stream_policy):
    anchors = O.tile_anchors(O.base_anchors(32, (1, 2, .5)), 6, 6, 8)
    gt = np.full((3, 5, 4), -1, F)
    lab = np.full((3, 5), -1, np.int32)
    gt[1, 0] = anchors[10]
    lab[1, 0] = 97                               # >= num_classes: tf.one_hot gives a zero row
    gt[2, 3] = anchors[20]                       # padding first, the match behind it
    lab[2, 3] = 5
    got = check(anchors, (48, 48), gt, lab, 80)
    assert (cpu(got['state'])[0] == -1).all() and (cpu(got['match_idx'])[0] == 0).all()
    row = cpu(got['labels'])[1, 10]
    assert row[:80].sum() == 0 and row[80] == 1
    row = cpu(got['labels'])[2, 20]
    assert row[5] == 1 and row[:80].sum() == 1 and row[80] == 1


def test_degenerate_boxes(stream_policy):
    """Inverted (negative-area) and zero-area anchors / GT rows, far-away rows, leading padding: the exact loop of
    the sweep's matcher and the culling matcher of the tile kernel must agree with the oracle and each other."""
    rng = np.random.default_rng(77)
    anchors = O.generate_anchors(256)
    inv = anchors.copy()
    inv[::7] = inv[::7][:, [2, 1, 0, 3]]             # x0 > x1: negative width
    inv[5::11, 3] = inv[5::11, 1]                    # zero height
    G = 40
    gts = [workloads.gt_boxes(rng, 256, G - 8, 80, pad_to=G) for _ in range(4)]
    gt = np.stack([g[0] for g in gts])
    lab = np.stack([g[1] for g in gts])
    gt[0, [0, 1]] = -1
    lab[0, [0, 1]] = -1
    gt[1, 3] = gt[1, 3][[2, 1, 0, 3]]
    gt[1, 4, 2] = gt[1, 4, 0]
    gt[2, :G - 8] = [1000, 1000, 1010, 1010]
    gt[3, :G - 8, :] += 4000
    for a in (anchors, inv):
        check(a, (256, 256), gt, lab, 80)


def test_boxes_whose_areas_overflow(stream_policy):
    """Finite coordinates of 1e19 .. 3e38: areas overflow to +inf, unions to inf - inf = NaN, a zero-width anchor of
    infinite height has area inf * 0 = NaN, offsets divide by inf.  The unit is built with NaNs honoured and the matcher
    sends every wave or row with a non-finite area through its exact loop, where a NaN overlap never wins (the rule of
    Eigen's reducers, oracle/odbox_oracle.c): the sweep, the 256-anchor tiles, the streaming tiles and the oracle must
    agree on every output."""
    rng = np.random.default_rng(1020)
    anchors = O.generate_anchors(256)
    N = anchors.shape[0]
    huge = anchors.copy()
    huge[3::97] = [-1e20, -1e20, 1e20, 1e20]                 # area inf
    huge[5::101] = [0, 0, 3e38, 3e38]                        # sides finite, area inf
    huge[7::103] = [5, -3e38, 5, 3e38]                       # width 0, height inf: area NaN
    huge[11::107] = [-3e38, 10, 3e38, 20]                    # width inf
    huge[13::109] = [1e19, 1e19, 3e19, 3e19]                 # far away and huge
    G = 24
    gts = [workloads.gt_boxes(rng, 256, G - 4, 80, pad_to=G) for _ in range(5)]
    gt = np.stack([g[0] for g in gts])
    lab = np.stack([g[1] for g in gts])
    gt[0, 0] = [-1e20, -1e20, 1e20, 1e20]                    # NaN overlap with the huge anchors in the FIRST row
    gt[1, 5] = [0, 0, 3e38, 3e38]
    gt[1, 6] = [-1e20, -1e20, 1e20, 1e20]
    gt[2, :G - 4] = [-1e20, -1e20, 1e20, 1e20]               # every valid row: NaN or 0 for everybody
    gt[3, 2] = [5, -3e38, 5, 3e38]                           # a NaN-area row
    gt[3, 3] = [100, 100, 3e38, 120]
    gt[4, 1] = [1e19, 1e19, 3e19, 3e19]
    with np.errstate(over='ignore', invalid='ignore'):
        for a in (huge, anchors):
            check(a, (256, 256), gt, lab, 80)
        # the streaming tiles (a GT list too long for the sweep) and, under 'cache', the 256-anchor tiles again
        gt65 = np.concatenate([gt, np.full((5, 65 - G, 4), -1, F)], 1)
        lab65 = np.concatenate([lab, np.full((5, 65 - G), -1, np.int32)], 1)
        gt65[:, 64] = [-1e20, 0, 1e20, 3e38]
        lab65[:, 64] = 7
        assert A.match_encode_plan(5, N, 65, 80) == 'tiles_stream'
        got = A.match_encode(huge, (256, 256), gt65, lab65, 80, dense=True, compact=True)
        ref = c_oracle.match_encode(huge, (256, 256), gt65, lab65, 80, 0.4, 0.5, threads=8)
        for k in ('match_idx', 'state', 'max_iou', 'labels'):
            np.testing.assert_array_equal(cpu(got[k]), ref[k])
        np.testing.assert_allclose(cpu(got['regression']), ref['regression'], rtol=1e-5, atol=0, equal_nan=True)


def test_ties_and_near_ties(stream_policy):
    """Rows that tie exactly and rows a few ulps apart: the sweep's exact divisions against the tile kernel's
    division-free loop with its exact fallback, and the oracle."""
    rng = np.random.default_rng(2024)
    anchors = O.generate_anchors(256)
    B, G = 4, 24
    gt = np.empty((B, G, 4), F)
    for b in range(B):
        base = workloads.gt_boxes(rng, 256, G // 4, 80)[0]
        rows = []
        for r in base:
            up = r.copy()
            up[2] = np.nextafter(up[2], F(1e9))
            dn = r.copy()
            dn[1] = np.nextafter(dn[1], F(-1e9))
            rows += [r, r.copy(), up, dn] if b % 2 == 0 else [up, dn, r, r.copy()]
        gt[b] = np.stack(rows)
    lab = rng.integers(0, 80, (B, G)).astype(np.int32)
    check(anchors, (256, 256), gt, lab, 80)


def test_cfg1_and_cfg2_full_size(stream_policy):
    anchors = O.generate_anchors(512)
    gt, lab = workloads.match_inputs(1)
    got = check(anchors, (512, 512), gt, lab, 80)
    assert (cpu(got['state']) == 1).sum() > 0
    anchors = O.yolo_shaped_anchors(416)
    gt, lab = workloads.match_inputs(2)
    check(anchors, (416, 416), gt, lab, 80)


def test_dense_only_and_compact_only_calls(stream_policy):
    """NULL compact outputs / NULL regression: the sweep writes what is asked for and nothing else."""
    lib = _lib.load()
    dev = torch.device('cuda')
    rng = np.random.default_rng(9)
    B, N, G, C = 3, 2001, 9, 80
    anchors, gt, lab = random_case(rng, B, N, G, C)
    a, g_, l_ = (torch.as_tensor(x).to(dev) for x in (anchors, gt, lab))
    ref = c_oracle.match_encode(anchors, (256, 256), gt, lab, C, threads=4)
    out = torch.full((B * N * (C + 1) + 64,), 7.0, device=dev)
    view = out[32:32 + B * N * (C + 1)]                    # 128 bytes into the allocation: still 16-byte aligned
    rc = lib.odbox_match_encode(a.data_ptr(), g_.data_ptr(), l_.data_ptr(), B, N, G, C, 256.0, 256.0, 0.4, 0.5,
                                None, view.data_ptr(), None, None, None, None, _lib.stream_ptr())
    assert rc == 0, lib.odbox_last_error()
    np.testing.assert_array_equal(cpu(view).reshape(B, N, C + 1), ref['labels'])
    assert float(out[:32].min()) == 7.0 and float(out[32 + B * N * (C + 1):].min()) == 7.0
    # labels 4 bytes off a 16-byte boundary: not the sweep, same bits
    assert A.match_encode_plan(B, N, G, C, labels_aligned16=False) == 'tiles_stream'
    view = out[33:33 + B * N * (C + 1)]
    rc = lib.odbox_match_encode(a.data_ptr(), g_.data_ptr(), l_.data_ptr(), B, N, G, C, 256.0, 256.0, 0.4, 0.5,
                                None, view.data_ptr(), None, None, None, None, _lib.stream_ptr())
    assert rc == 0, lib.odbox_last_error()
    np.testing.assert_array_equal(cpu(view).reshape(B, N, C + 1), ref['labels'])


def test_repeated_calls_into_dirty_buffers(stream_policy):
    """Every element is written on every call (zeros by the store waves, patches behind them): a buffer full of
    ones and a second, different batch into the same buffer both come out exact."""
    lib = _lib.load()
    dev = torch.device('cuda')
    anchors = O.yolo_shaped_anchors(416)
    N, C, B = anchors.shape[0], 80, 8
    a = torch.as_tensor(anchors).to(dev)
    lab_out = torch.ones((B, N, C + 1), device=dev)
    reg_out = torch.ones((B, N, 5), device=dev)
    for seed in (1, 2, 3):
        gt, lab = workloads.match_inputs(2, batch=B, seed=seed)
        g_, l_ = torch.as_tensor(gt).to(dev), torch.as_tensor(lab).to(dev)
        for _ in range(3):
            rc = lib.odbox_match_encode(a.data_ptr(), g_.data_ptr(), l_.data_ptr(), B, N, gt.shape[1], C, 416.0, 416.0, 0.4,
                                        0.5, reg_out.data_ptr(), lab_out.data_ptr(), None, None, None, None, _lib.stream_ptr())
            assert rc == 0, lib.odbox_last_error()
        ref = c_oracle.match_encode(anchors, (416, 416), gt, lab, C, threads=8)
        np.testing.assert_array_equal(cpu(lab_out), ref['labels'])
        np.testing.assert_array_equal(cpu(reg_out)[..., 4], ref['regression'][..., 4])


def test_auto_policy_is_a_function_of_the_shape_alone():
    """AUTO (the default) keeps no state: the same call launches the same kernel whether its buffer is fresh, was
    written a moment ago, in a rotation or long ago -- the sweep wherever it applies, as for a cold buffer -- and
    odbox_match_encode_plan says which beforehand.  'cache' is how a caller asks for the other one.  The bits are the
    same either way."""
    import time
    lib = _lib.load()
    dev = torch.device('cuda')
    assert A.set_output_policy('auto') == 'auto'
    anchors = O.yolo_shaped_anchors(416)
    N, C, B = anchors.shape[0], 80, 24                                  # 83 MB of labels per buffer
    gt, lab = workloads.match_inputs(2, batch=B)
    a, g_, l_ = (torch.as_tensor(x).to(dev) for x in (anchors, gt, lab))
    bufs = [torch.empty((B, N, C + 1), device=dev) for _ in range(3)]
    regs = [torch.empty((B, N, 5), device=dev) for _ in range(3)]

    def call(k):
        rc = lib.odbox_match_encode(a.data_ptr(), g_.data_ptr(), l_.data_ptr(), B, N, gt.shape[1], C, 416.0, 416.0, 0.4, 0.5,
                                    regs[k].data_ptr(), bufs[k].data_ptr(), None, None, None, None, _lib.stream_ptr())
        assert rc == 0, lib.odbox_last_error()
        return A.match_encode_last_plan()

    assert A.match_encode_plan(B, N, gt.shape[1], C) == 'sweep'
    assert [call(k) for k in (0, 0, 0, 1, 0, 1, 2, 2)] == ['sweep'] * 8  # fresh, in place, in rotation: one answer
    torch.cuda.synchronize()
    time.sleep(0.02)
    assert call(2) == 'sweep'
    assert A.match_encode_plan(B, N, 65, C) == 'tiles'                  # a shape the sweep does not serve, below the size rule
    prev = A.set_output_policy('cache')
    try:
        assert A.match_encode_plan(B, N, gt.shape[1], C) == 'tiles' and call(1) == 'tiles'
    finally:
        A.set_output_policy(prev)
    torch.cuda.synchronize()
    ref = c_oracle.match_encode(anchors, (416, 416), gt, lab, C, threads=8)
    for k in range(3):
        np.testing.assert_array_equal(cpu(bufs[k]), ref['labels'])
        np.testing.assert_array_equal(cpu(regs[k])[..., 4], ref['regression'][..., 4])


def test_padding_layouts_and_rows_beyond_the_lds_share(stream_policy):
    """The sweep's matcher never visits the padding behind an image's last valid row and keeps only the first
    3 072 / B rows of every image in LDS.  Padded to 64 like the reference's batches, padding in FRONT of and BETWEEN
    valid rows, all-padded images, and batches large enough that valid rows lie beyond the LDS share (B = 200: 15
    rows per image; B = 3 100: none) must all give the oracle's argmax."""
    rng = np.random.default_rng(4)
    anchors = O.tile_anchors(O.base_anchors(32, (1, 2, .5)), 7, 5, 16)      # 315 anchors
    for B, G, C in ((6, 64, 80), (200, 40, 70), (3100, 3, 68)):
        gts = [workloads.gt_boxes(rng, 112, G, C) for _ in range(B)]
        gt = np.stack([g[0] for g in gts])
        lab = np.stack([g[1] for g in gts])
        for b in range(B):
            kind = b % 6
            if kind == 0:
                gt[b, rng.integers(1, G):] = -1                    # the usual case: a valid prefix, then padding
            elif kind == 1 and G > 4:
                gt[b, :G - 2] = -1                                 # padding first, two boxes at the very end
            elif kind == 2 and G > 4:
                gt[b, 1::2] = -1                                   # every other row padded
            elif kind == 3:
                gt[b] = -1                                         # nothing but padding
            elif kind == 4 and G > 8:
                gt[b, 2:G - 3] = -1                                # a hole
            lab[b][gt[b, :, 2] < 0] = -1
        check(anchors, (112, 80), gt, lab, C)


@pytest.mark.parametrize('policy', ['stream', 'auto'])
def test_sweep_replays_from_a_hip_graph(policy):
    """The sweep allocates nothing and reads everything it decides on from device memory (the per-image bounds are
    recomputed by every launch): captured once, replayed on batches with other boxes and other amounts of padding.
    Under the default policy too: AUTO is a function of the shape, so the captured kernel is the one
    `match_encode_plan` names, not a guess frozen at capture time."""
    prev = A.set_output_policy(policy)
    try:
        _sweep_graph_case()
    finally:
        A.set_output_policy(prev)


def _sweep_graph_case():
    lib = _lib.load()
    dev = torch.device('cuda')
    anchors_np = O.generate_anchors(256)
    N, C, B, G = anchors_np.shape[0], 80, 5, 64
    anchors = torch.as_tensor(anchors_np).to(dev)
    assert A.match_encode_plan(B, N, G, C) == 'sweep'                     # what the capture below will hold

    def inputs(seed, valid):
        r = np.random.default_rng(seed)
        gts = [workloads.gt_boxes(r, 256, valid, C, pad_to=G) for _ in range(B)]
        return np.stack([g[0] for g in gts]), np.stack([g[1] for g in gts])

    gt_np, lab_np = inputs(1, 20)
    gt, lab = torch.as_tensor(gt_np).to(dev), torch.as_tensor(lab_np).to(dev)
    out_reg = torch.empty((B, N, 5), device=dev)
    out_lab = torch.empty((B, N, C + 1), device=dev)
    idx = torch.empty((B, N), dtype=torch.int32, device=dev)

    def enqueue():
        rc = lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, 256.0, 256.0, 0.4, 0.5,
                                    out_reg.data_ptr(), out_lab.data_ptr(), idx.data_ptr(), None, None, None, _lib.stream_ptr())
        assert rc == 0, lib.odbox_last_error()
        assert A.match_encode_last_plan() == 'sweep'

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        enqueue()
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        enqueue()
    for seed, valid in ((2, 3), (3, 64), (4, 41)):
        g2, l2 = inputs(seed, valid)
        gt.copy_(torch.as_tensor(g2)), lab.copy_(torch.as_tensor(l2))
        out_lab.fill_(7.0), out_reg.fill_(7.0), idx.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        ref = c_oracle.match_encode(anchors_np, (256, 256), g2, l2, C, threads=4)
        np.testing.assert_array_equal(cpu(out_lab), ref['labels'])
        np.testing.assert_array_equal(cpu(idx), ref['match_idx'])
        np.testing.assert_array_equal(cpu(out_reg)[..., 4], ref['regression'][..., 4])
