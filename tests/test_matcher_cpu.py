"""CPU: bit-exact index parity of uninext_amd.matcher.HungarianMatcherVL with the reference matcher.

Fixtures (tests/golden/matcher_*.npz) were produced by running the reference's own matcher.py on seeded CPU inputs
(tests/golden/make_matcher_golden.py); BASELINE.json asks for integer equality, so every comparison is array_equal.
"""
import numpy as np
import pytest
import torch

from golden_util import load_golden, matcher_names
from uninext_amd.matcher import HungarianMatcherVL, box_iou, generalized_box_iou

NAMES = matcher_names()


def _case(name, device="cpu"):
    g = load_golden(name)
    bs = int(g["bs"])
    outputs = {"pred_logits": torch.from_numpy(g["pred_logits"]).to(device),
               "pred_boxes": torch.from_numpy(g["pred_boxes"]).to(device)}
    targets = [{"boxes": torch.from_numpy(g[f"tgt_boxes_{b}"]).to(device),
                "positive_map": torch.from_numpy(g[f"tgt_posmap_{b}"]).to(device)} for b in range(bs)]
    return g, bs, outputs, targets


def test_fixtures_present():
    assert len(NAMES) >= 6


@pytest.mark.parametrize("name", NAMES)
def test_hungarian_indices_bit_exact(name):
    g, bs, outputs, targets = _case(name)
    result = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2).forward(outputs, targets)
    assert len(result) == bs
    for b, (i, j) in enumerate(result):
        assert i.dtype == torch.int64 and j.dtype == torch.int64
        assert np.array_equal(i.numpy(), g[f"hung_i_{b}"])
        assert np.array_equal(j.numpy(), g[f"hung_j_{b}"])
        assert len(i) == min(outputs["pred_logits"].shape[1], len(targets[b]["boxes"]))


@pytest.mark.parametrize("name", [n for n in NAMES if "encoder" not in n])
def test_ota_indices_bit_exact(name):
    g, bs, outputs, targets = _case(name)
    indices, matched = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2).forward_ota(outputs, targets)
    for b in range(bs):
        q, gt = indices[b]
        assert q.dtype == torch.int64 and gt.dtype == torch.int64
        assert np.array_equal(q.numpy(), g[f"ota_q_{b}"])
        assert np.array_equal(gt.numpy(), g[f"ota_g_{b}"])
        m = matched[b].numpy() if torch.is_tensor(matched[b]) else np.asarray(matched[b], dtype=np.int64)
        assert np.array_equal(m, g[f"ota_matched_{b}"])
        if len(targets[b]["boxes"]) and "conflicts" not in name:
            assert set(gt.tolist()) == set(range(len(targets[b]["boxes"])))   # every gt got at least one query
        # (matcher_ota_conflicts: clustered, partly identical targets -- the reference's repair loop leaves queries that hold
        # two targets, of which `matching[selected].max(1)[1]` reports the first: fewer distinct targets than G, by design)
        assert len(set(q.tolist())) == len(q)                                # a query is listed once


def _focal_table(logits):
    from uninext_amd.matcher import FOCAL_ALPHA, FOCAL_GAMMA
    prob = logits.sigmoid()
    neg = (1 - FOCAL_ALPHA) * (prob ** FOCAL_GAMMA) * (-(1 - prob + 1e-8).log())
    pos = FOCAL_ALPHA * ((1 - prob) ** FOCAL_GAMMA) * (-(prob + 1e-8).log())
    return pos - neg


@pytest.mark.parametrize("name", [n for n in NAMES if "encoder" not in n])
def test_ota_oracle_is_pinned_to_the_reference_fixtures(name):
    """oracle/ota_oracle.py -- the float32 numpy restatement of matcher.py:313-447 the HIP kernels of include/ota_hip.h are
    held to on the GPU -- returns the reference matcher's own integers on every fixture, incl. exact ties, an image without
    targets and the conflict-heavy case whose repair loop runs."""
    from oracle import ota_oracle
    g, bs, outputs, targets = _case(name)
    table = _focal_table(outputs["pred_logits"]).numpy()
    for b in range(bs):
        if len(targets[b]["boxes"]) == 0:
            continue
        cost, iou, flags = ota_oracle.cost_terms(table[b], outputs["pred_boxes"][b].numpy(), targets[b]["boxes"].numpy(),
                                                 targets[b]["positive_map"].numpy())
        # the cost matrix itself against the mirror's composition (same torch CPU operations as the reference)
        m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
        want_cost, want_iou, _ = m.compute_cost(b, outputs["pred_boxes"], outputs["pred_logits"].sigmoid(), targets, 1)
        assert np.array_equal(iou, want_iou.numpy())
        # bitwise for targets of one or two tokens; three tokens: PyTorch's CPU mean divides by 3 where its GPU mean (the
        # kernels' and the oracle's rule) multiplies by fl(1 / 3) -- one unit in the last place of a cost near 100
        full = cost + np.where((flags & 1).any(1), 0, 10000.0).astype(np.float32)[:, None]
        ntok = targets[b]["positive_map"].sum(1).numpy()
        assert np.array_equal(full[:, ntok <= 2], want_cost.numpy()[:, ntok <= 2])
        assert float(np.abs(full - want_cost.numpy()).max()) <= 7.63e-6 * max(1.0, float(np.abs(full).max()) / 64.0)
        sel, gt, matched, M, status = ota_oracle.dynamic_k(cost, iou, flags)
        assert status == 0
        assert np.array_equal(sel, g[f"ota_q_{b}"]) and np.array_equal(gt, g[f"ota_g_{b}"])
        assert np.array_equal(matched, g[f"ota_matched_{b}"])


@pytest.mark.parametrize("seed", range(12))
def test_ota_oracle_equals_the_reference_loop_on_conflict_heavy_inputs(seed):
    """Random clustered / duplicated targets (the repair loop runs in most of them): the oracle against the mirror's
    composition with the reference's per-target top-k loop."""
    from oracle import ota_oracle
    g = torch.Generator().manual_seed(500 + seed)
    Q, T, G = 150, 16, int(torch.randint(2, 60, (1,), generator=g))
    logits = torch.randn(1, Q, T, generator=g) * 2
    boxes = torch.cat([torch.rand(1, Q, 2, generator=g), 0.03 + 0.3 * torch.rand(1, Q, 2, generator=g) ** 2], -1)
    c = 0.4 + 0.2 * torch.rand(G, 2, generator=g)
    tb = torch.cat([c, 0.1 + 0.2 * torch.rand(G, 2, generator=g)], -1)
    if seed % 3 == 0 and G > 4:
        tb[1::4] = tb[0::4][:len(tb[1::4])]
    pm = torch.zeros(G, T, dtype=torch.bool)
    pm[torch.arange(G), torch.randint(0, T, (G,), generator=g)] = True
    if seed % 2:
        pm[torch.arange(G), torch.randint(0, T, (G,), generator=g)] = True       # two or three tokens per target
        pm[torch.arange(G), torch.randint(0, T, (G,), generator=g)] = True
    targets = [{"boxes": tb, "positive_map": pm}]
    m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    m.batched_topk = False
    (ref_idx,), (ref_matched,) = m.forward_ota({"pred_logits": logits, "pred_boxes": boxes}, targets)
    cost, iou, flags = ota_oracle.cost_terms(_focal_table(logits)[0].numpy(), boxes[0].numpy(), tb.numpy(), pm.numpy())
    sel, gt, matched, M, status = ota_oracle.dynamic_k(cost, iou, flags)
    assert status == 0
    assert np.array_equal(sel, ref_idx[0].numpy()) and np.array_equal(gt, ref_idx[1].numpy())
    assert np.array_equal(matched, ref_matched.numpy())


@pytest.mark.parametrize("seed", range(4))
def test_ota_oracle_follows_pytorch_on_nan_ious_of_zero_area_boxes(seed):
    """a zero-area query on a zero-area target passes the reference's box assert (x1 >= x0 holds) and has IoU 0 / 0.
    torch.topk ranks that NaN LARGEST (the column's k becomes int(NaN) clamped to 1), torch.min / argmin PROPAGATE it (the
    NaN's row is "the cheapest", matched or not): the oracle -- and with it the kernels of include/ota_hip.h -- do the same."""
    from oracle import ota_oracle
    g = torch.Generator().manual_seed(600 + seed)
    Q, T, G = 60, 8, 4 + seed
    logits = torch.randn(1, Q, T, generator=g)
    boxes = torch.cat([torch.rand(1, Q, 2, generator=g), 0.05 + 0.3 * torch.rand(1, Q, 2, generator=g)], -1)
    tb = torch.cat([0.3 + 0.4 * torch.rand(G, 2, generator=g), 0.1 + 0.3 * torch.rand(G, 2, generator=g)], -1)
    tb[2, 2:] = 0.0
    boxes[0, 9 + seed] = tb[2]
    pm = torch.zeros(G, T, dtype=torch.bool)
    pm[torch.arange(G), torch.arange(G) % T] = True
    m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    m.batched_topk = False
    (ref_idx,), (ref_matched,) = m.forward_ota({"pred_logits": logits, "pred_boxes": boxes}, [{"boxes": tb, "positive_map": pm}])
    cost, iou, flags = ota_oracle.cost_terms(_focal_table(logits)[0].numpy(), boxes[0].numpy(), tb.numpy(), pm.numpy())
    assert np.isnan(iou).sum() == 1 and not (flags & 2).any()
    sel, gt, matched, M, status = ota_oracle.dynamic_k(cost, iou, flags)
    assert status == 0
    assert np.array_equal(sel, ref_idx[0].numpy()) and np.array_equal(gt, ref_idx[1].numpy())
    assert np.array_equal(matched, ref_matched.numpy())


def test_ota_oracle_flags_degenerate_boxes_like_the_reference_assert():
    """the reference's generalized_box_iou asserts `(boxes[:, 2:] >= boxes[:, :2]).all()` on the predicted and
    on the target boxes (util/box_ops.py:76-77) -- a negative width / height or a NaN aborts the step.  The device path
    (include/ota_hip.h) reports it as bit 1 of the pair flags and status bit 2 (4) of the image; the oracle mirrors that, and
    the composition path raises the AssertionError itself."""
    from oracle import ota_oracle
    g = torch.Generator().manual_seed(77)
    Q, T, G = 40, 8, 3
    logits = torch.randn(1, Q, T, generator=g)
    boxes = torch.cat([torch.rand(1, Q, 2, generator=g), 0.05 + 0.2 * torch.rand(1, Q, 2, generator=g)], -1)
    tb = torch.cat([0.3 + 0.4 * torch.rand(G, 2, generator=g), 0.1 + 0.2 * torch.rand(G, 2, generator=g)], -1)
    pm = torch.zeros(G, T, dtype=torch.bool)
    pm[torch.arange(G), torch.arange(G)] = True
    table = _focal_table(logits)[0].numpy()
    cost, iou, flags = ota_oracle.cost_terms(table, boxes[0].numpy(), tb.numpy(), pm.numpy())
    assert not (flags & 2).any() and ota_oracle.dynamic_k(cost, iou, flags)[4] == 0
    for what in ("negative width", "nan", "negative target height"):
        b2, t2 = boxes.clone(), tb.clone()
        if what == "negative width":
            b2[0, 5, 2] = -0.1
        elif what == "nan":
            b2[0, 7, 1] = float("nan")
        else:
            t2[1, 3] = -0.05
        cost, iou, flags = ota_oracle.cost_terms(table, b2[0].numpy(), t2.numpy(), pm.numpy())
        assert (flags & 2).any()
        if what != "negative target height":
            row = 5 if what == "negative width" else 7
            assert (flags[row] & 2).all() and not (np.delete(flags, row, 0) & 2).any()
        assert ota_oracle.dynamic_k(cost, iou, flags)[4] & 4
        m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
        with pytest.raises(AssertionError):                       # the composition = the reference's own behaviour
            m.forward_ota({"pred_logits": logits, "pred_boxes": b2}, [{"boxes": t2, "positive_map": pm}])


@pytest.mark.parametrize("seed", range(8))
def test_batched_dynamic_k_equals_reference_loop(seed):
    """The sync-free top-10 + rank-mask selection picks exactly what the reference's per-gt topk loop picks, also with
    many, clustered and duplicated ground-truth boxes (repair loop exercised)."""
    g = torch.Generator().manual_seed(100 + seed)
    bs, Q, T = 2, 300, 16
    logits = torch.randn(bs, Q, T, generator=g) * 2
    centers = torch.rand(bs, Q, 2, generator=g)
    boxes = torch.cat([centers, 0.03 + 0.3 * torch.rand(bs, Q, 2, generator=g) ** 2], -1)
    targets = []
    for b in range(bs):
        G = int(torch.randint(1, 40, (1,), generator=g))
        c = 0.3 + 0.4 * torch.rand(G, 2, generator=g) if seed % 2 else torch.rand(G, 2, generator=g)
        tb = torch.cat([c, 0.05 + 0.25 * torch.rand(G, 2, generator=g)], -1)
        if G > 3:
            tb[1] = tb[0]                                   # duplicated gt: contested queries
        pm = torch.zeros(G, T, dtype=torch.bool)
        pm[torch.arange(G), torch.randint(0, T, (G,), generator=g)] = True
        targets.append({"boxes": tb, "positive_map": pm, "labels": torch.zeros(G, dtype=torch.int64),
                        "image_size": torch.tensor([800.0, 1333.0, 800.0, 1333.0])})
    outputs = {"pred_logits": logits, "pred_boxes": boxes}
    m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    fast = m.forward_ota(outputs, targets)
    HungarianMatcherVL.batched_topk = False
    try:
        ref = m.forward_ota(outputs, targets)
    finally:
        HungarianMatcherVL.batched_topk = True
    for b in range(bs):
        assert torch.equal(fast[0][b][0], ref[0][b][0]) and torch.equal(fast[0][b][1], ref[0][b][1])
        assert torch.equal(torch.as_tensor(fast[1][b]), torch.as_tensor(ref[1][b]))


def test_box_helpers_against_closed_forms():
    a = torch.tensor([[0.0, 0.0, 2.0, 2.0], [1.0, 1.0, 3.0, 3.0]])
    b = torch.tensor([[1.0, 1.0, 2.0, 2.0], [4.0, 4.0, 5.0, 5.0]])
    iou = box_iou(a, b)
    assert torch.allclose(iou, torch.tensor([[0.25, 0.0], [0.25, 0.0]]))
    giou = generalized_box_iou(a, b)
    assert torch.allclose(giou[0, 1], torch.tensor(0.0 - (25.0 - 5.0) / 25.0), atol=1e-6)
    with pytest.raises(AssertionError):
        generalized_box_iou(torch.tensor([[1.0, 1.0, 0.0, 0.0]]), b)


def test_constructor_rejects_all_zero_costs():
    with pytest.raises(AssertionError):
        HungarianMatcherVL(0, 0, 0, 0)
