"""GPU: the matcher mirror with predictions/targets resident on the MI355X (cost terms run on the device, the
assignment on the host as in the reference) must return the same integer indices as the reference-minted fixtures."""
import numpy as np
import pytest
import torch

from golden_util import matcher_names
from test_matcher_cpu import _case
from uninext_amd.matcher import HungarianMatcherVL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", matcher_names())
def test_hungarian_on_device_matches_reference(name, fused):
    """fused: the cost matrix from ONE kernel (include/matcher_cost_hip.h); otherwise the PyTorch composition."""
    g, bs, outputs, targets = _case(name, device="cuda:0")
    matcher = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    matcher.fused_cost = fused
    result = matcher.forward(outputs, targets)
    for b, (i, j) in enumerate(result):
        assert np.array_equal(i.numpy(), g[f"hung_i_{b}"]) and np.array_equal(j.numpy(), g[f"hung_j_{b}"])


def test_default_hungarian_cost_is_bitwise_the_reference_composition():
    """the contract of this row is bit-exact indices.  The DEFAULT cost matrix is the reference's own chain of
    PyTorch operations on the device (matcher.py:476-498) -- bitwise equal to it on every fixture and on 200 seeded batches with
    multi-token targets -- and on constructed near-ties (two queries whose costs for one target differ by one unit in the last
    place, and exact ties) the device assignment returns SciPy's indices on that very matrix."""
    from scipy.optimize import linear_sum_assignment
    from uninext_amd.matcher import box_cxcywh_to_xyxy, focal_token_cost, generalized_box_iou
    assert HungarianMatcherVL.fused_cost is False
    matcher = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)

    def composition(logits, boxes, tgt_map, tgt_boxes):
        cls = focal_token_cost(logits.sigmoid(), tgt_map)
        l1 = torch.cdist(boxes, tgt_boxes, p=1)
        gi = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt_boxes))
        return 5 * l1 + 2 * cls + 2 * gi

    cases = []
    for name in matcher_names():
        g, bs, outputs, targets = _case(name, device="cuda:0")
        if sum(len(t["boxes"]) for t in targets):
            cases.append((outputs, targets))
    gen = torch.Generator().manual_seed(77)
    for k in range(200):
        Q, T, G = 60 + k % 7, 32, 3 + k % 5
        pm = torch.zeros(G, T, dtype=torch.bool)
        for t in range(G):
            n_tok = 1 + int(torch.randint(0, 9, (1,), generator=gen))          # 1..9 tokens: PyTorch's mean(-1) in every regime
            pm[t, torch.randperm(T, generator=gen)[:n_tok]] = True
        cxcy = torch.rand(G, 2, generator=gen) * 0.6 + 0.2
        tb = torch.cat([cxcy, torch.rand(G, 2, generator=gen) * 0.2 + 0.05], 1)
        outputs = {"pred_logits": torch.randn(1, Q, T, generator=gen).cuda(),
                   "pred_boxes": torch.cat([torch.rand(1, Q, 2, generator=gen) * 0.6 + 0.2, torch.rand(1, Q, 2, generator=gen) * 0.2 + 0.05], 2).cuda()}
        cases.append((outputs, [{"boxes": tb.cuda(), "positive_map": pm.cuda()}]))
    for outputs, targets in cases:
        logits, boxes = outputs["pred_logits"].flatten(0, 1), outputs["pred_boxes"].flatten(0, 1)
        tgt_map = torch.cat([t["positive_map"] for t in targets])
        tgt_boxes = torch.cat([t["boxes"] for t in targets])
        assert torch.equal(matcher.cost_matrix(logits, boxes, tgt_map, tgt_boxes), composition(logits, boxes, tgt_map, tgt_boxes))
    # near-ties: duplicate a query, then move its copy's cost for one target by exactly one ulp either way (and not at all)
    outputs, targets = cases[-1]
    for shift in (-1, 0, 1):
        logits, boxes = outputs["pred_logits"][0].clone(), outputs["pred_boxes"][0].clone()
        logits[1], boxes[1] = logits[0], boxes[0]
        cost = composition(logits, boxes, targets[0]["positive_map"], targets[0]["boxes"])
        bits = cost.view(torch.int32)
        bits[1, 0] += shift * (1 if float(cost[1, 0]) >= 0 else -1)
        from uninext_amd import ext
        (i, j), = ext.lsap_batch([cost], check=True)
        ri, rj = linear_sum_assignment(cost.cpu().numpy())
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(j.cpu().numpy(), rj), shift


OTA_NAMES = [n for n in matcher_names() if "encoder" not in n]


@pytest.mark.parametrize("device_ota", [True, False])
@pytest.mark.parametrize("name", OTA_NAMES)
def test_ota_on_device_matches_reference(name, device_ota):
    """device_ota: the two HIP kernels of include/ota_hip.h; otherwise the PyTorch composition.  Both return the
    reference matcher's integers on every fixture -- exact ties, an image without targets, the conflict-heavy case."""
    g, bs, outputs, targets = _case(name, device="cuda:0")
    matcher = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    matcher.device_ota = device_ota
    indices, matched = matcher.forward_ota(outputs, targets)
    for b in range(bs):
        assert indices[b][0].is_cuda or len(targets[b]["boxes"]) == 0 or indices[b][0].device.type == "cuda"
        assert np.array_equal(indices[b][0].cpu().numpy(), g[f"ota_q_{b}"])
        assert np.array_equal(indices[b][1].cpu().numpy(), g[f"ota_g_{b}"])
        m = matched[b].cpu().numpy() if torch.is_tensor(matched[b]) else np.asarray(matched[b], dtype=np.int64)
        assert np.array_equal(m, g[f"ota_matched_{b}"])


def test_ota_device_path_never_synchronises_before_its_one_copy():
    """zero host syncs until the final index copy.  Everything up to the copy of the per-image counts runs
    under torch.cuda.set_sync_debug_mode("error") -- any synchronising PyTorch call (nonzero, .item(), boolean-mask
    indexing, .cpu(), `if tensor:`) raises there; the PyTorch composition, for comparison, does raise."""
    g, bs, outputs, targets = _case("matcher_q900_t256", device="cuda:0")
    matcher = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    prob = outputs["pred_logits"].sigmoid()
    matcher.ota_device_launch(prob, outputs["pred_boxes"], targets)      # warm-up: library load, allocator
    torch.cuda.synchronize()
    old = torch.cuda.get_sync_debug_mode()
    torch.cuda.set_sync_debug_mode("error")
    try:
        sel_q, sel_g, matched, count, status, sizes = matcher.ota_device_launch(prob, outputs["pred_boxes"], targets)
        matcher.device_ota = False
        with pytest.raises(RuntimeError):
            matcher.forward_ota(outputs, targets)                         # the composition synchronises (per-target masks, .any())
    finally:
        torch.cuda.set_sync_debug_mode(old)
    counts = count.cpu().tolist()                                         # the one copy
    assert status.cpu().tolist() == [0] * bs
    for b in range(bs):
        assert np.array_equal(sel_q[b, :counts[b]].cpu().numpy(), g[f"ota_q_{b}"])
        assert np.array_equal(sel_g[b, :counts[b]].cpu().numpy(), g[f"ota_g_{b}"])


@pytest.mark.parametrize("name", OTA_NAMES)
def test_ota_kernels_against_the_oracle_array_for_array(name):
    """ota_cost_hip_f32: cost, IoU and prior flags BITWISE the oracle's on the same focal table (the table is PyTorch's on
    the GPU, copied to the host for the oracle); ota_dynamic_k_hip: the final cost matrix (penalties applied in place), the
    0 / 1 matching matrix and all index outputs equal to the oracle's."""
    from oracle import ota_oracle
    from uninext_amd import ext
    from uninext_amd.matcher import FOCAL_ALPHA, FOCAL_GAMMA
    g, bs, outputs, targets = _case(name, device="cuda:0")
    prob = outputs["pred_logits"].sigmoid()
    neg = (1 - FOCAL_ALPHA) * (prob ** FOCAL_GAMMA) * (-(1 - prob + 1e-8).log())
    pos = FOCAL_ALPHA * ((1 - prob) ** FOCAL_GAMMA) * (-(prob + 1e-8).log())
    table = pos - neg
    sizes = [len(t["boxes"]) for t in targets]
    Q = prob.shape[1]
    tb = torch.cat([t["boxes"] for t in targets])
    pm = torch.cat([t["positive_map"] for t in targets])
    # the cost kernel alone (through the C ABI, like ext.ota_assign does)
    import ctypes
    lib = ext._lib.load()
    off = [0]
    for n in sizes:
        off.append(off[-1] + n)
    G = off[-1]
    cost = torch.empty(Q * G, device="cuda:0")
    iou = torch.empty(Q * G, device="cuda:0")
    flags = torch.empty(Q * G, dtype=torch.uint8, device="cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.ota_cost_hip_f32(table.contiguous().data_ptr(), outputs["pred_boxes"].contiguous().data_ptr(), tb.data_ptr(),
                              pm.view(torch.uint8).data_ptr(), (ctypes.c_int32 * (bs + 1))(*off), bs, Q, prob.shape[2],
                              cost.data_ptr(), iou.data_ptr(), flags.data_ptr(), stream)
    assert rc == 0
    sel_q, sel_g, matched, count, status = ext.ota_assign(table, outputs["pred_boxes"], tb, pm, sizes)
    counts = count.cpu().tolist()
    table_h = table.cpu().numpy()
    for b in range(bs):
        n = sizes[b]
        if n == 0:
            assert counts[b] == 0
            continue
        blk = slice(Q * off[b], Q * off[b + 1])
        o_cost, o_iou, o_flags = ota_oracle.cost_terms(table_h[b], outputs["pred_boxes"][b].cpu().numpy(), targets[b]["boxes"].cpu().numpy(),
                                                      targets[b]["positive_map"].cpu().numpy())
        k_cost, k_iou = cost[blk].view(Q, n).cpu().numpy(), iou[blk].view(Q, n).cpu().numpy()
        assert np.array_equal(k_cost.view(np.uint32), o_cost.view(np.uint32)), float(np.abs(k_cost - o_cost).max())
        assert np.array_equal(k_iou.view(np.uint32), o_iou.view(np.uint32))
        assert np.array_equal(flags[blk].view(Q, n).cpu().numpy(), o_flags)
        sel, gt, o_matched, M, st = ota_oracle.dynamic_k(o_cost, o_iou, o_flags)
        assert np.array_equal(sel_q[b, :counts[b]].cpu().numpy(), sel) and np.array_equal(sel_g[b, :counts[b]].cpu().numpy(), gt)
        assert np.array_equal(matched[off[b]:off[b + 1]].cpu().numpy(), o_matched)


@pytest.mark.parametrize("seed", range(10))
def test_ota_device_on_conflict_heavy_batches(seed):
    """Random batches of clustered / duplicated targets, 1..3 tokens per class name, an empty image in between, more
    queries than one pass of the workgroup (Q = 1500): device path == oracle == the reference's per-target loop."""
    from oracle import ota_oracle
    from uninext_amd.matcher import FOCAL_ALPHA, FOCAL_GAMMA
    g = torch.Generator().manual_seed(900 + seed)
    bs, T = 3, 24
    Q = 1500 if seed == 0 else 200
    logits = torch.randn(bs, Q, T, generator=g) * 2
    boxes = torch.cat([torch.rand(bs, Q, 2, generator=g), 0.03 + 0.3 * torch.rand(bs, Q, 2, generator=g) ** 2], -1)
    targets = []
    for b in range(bs):
        G = 0 if (b == 1 and seed % 2) else int(torch.randint(1, 70, (1,), generator=g))
        c = 0.4 + 0.2 * torch.rand(G, 2, generator=g)
        tb = torch.cat([c, 0.1 + 0.2 * torch.rand(G, 2, generator=g)], -1)
        if G > 4 and seed % 3 == 0:
            tb[1::4] = tb[0::4][:len(tb[1::4])]
        pm = torch.zeros(G, T, dtype=torch.bool)
        for _ in range(1 + seed % 3):
            pm[torch.arange(G), torch.randint(0, T, (G,), generator=g)] = True
        targets.append({"boxes": tb.cuda(), "positive_map": pm.cuda()})
    outputs = {"pred_logits": logits.cuda(), "pred_boxes": boxes.cuda()}
    m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    dev_idx, dev_matched = m.forward_ota(outputs, targets)
    prob = outputs["pred_logits"].sigmoid()
    neg = (1 - FOCAL_ALPHA) * (prob ** FOCAL_GAMMA) * (-(1 - prob + 1e-8).log())
    pos = FOCAL_ALPHA * ((1 - prob) ** FOCAL_GAMMA) * (-(prob + 1e-8).log())
    table = (pos - neg).cpu().numpy()
    m.device_ota, m.batched_topk = False, False
    loop_idx, loop_matched = m.forward_ota(outputs, targets)
    for b in range(bs):
        n = len(targets[b]["boxes"])
        if n == 0:
            assert dev_idx[b][0].numel() == 0 and dev_matched[b] == []
            continue
        cost, iou, flags = ota_oracle.cost_terms(table[b], boxes[b].numpy(), targets[b]["boxes"].cpu().numpy(), targets[b]["positive_map"].cpu().numpy())
        sel, gt, o_matched, M, st = ota_oracle.dynamic_k(cost, iou, flags)
        assert np.array_equal(dev_idx[b][0].cpu().numpy(), sel) and np.array_equal(dev_idx[b][1].cpu().numpy(), gt)
        assert np.array_equal(dev_matched[b].cpu().numpy(), o_matched)
        if seed % 3 != 0:      # without exactly duplicated targets the composition's GPU top-k has no ties to break its own way
            assert np.array_equal(dev_idx[b][0].cpu().numpy(), loop_idx[b][0].cpu().numpy())
            assert np.array_equal(dev_idx[b][1].cpu().numpy(), loop_idx[b][1].cpu().numpy())
            assert np.array_equal(dev_matched[b].cpu().numpy(), loop_matched[b].cpu().numpy())


def test_ota_device_follows_pytorch_on_nan_ious_of_zero_area_boxes():
    """tests/test_matcher_cpu.py::test_ota_oracle_follows_pytorch_on_nan_ious_of_zero_area_boxes, on the device: k = 1 for a
    column with a NaN IoU, the NaN's row wins every min / argmin."""
    from oracle import ota_oracle
    from uninext_amd.matcher import FOCAL_ALPHA, FOCAL_GAMMA
    for seed in range(4):
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
        outputs = {"pred_logits": logits.cuda(), "pred_boxes": boxes.cuda()}
        (idx,), (matched,) = m.forward_ota(outputs, [{"boxes": tb.cuda(), "positive_map": pm.cuda()}])
        prob = outputs["pred_logits"].sigmoid()
        table = (FOCAL_ALPHA * ((1 - prob) ** FOCAL_GAMMA) * (-(prob + 1e-8).log())
                 - (1 - FOCAL_ALPHA) * (prob ** FOCAL_GAMMA) * (-(1 - prob + 1e-8).log())).cpu().numpy()
        cost, iou, flags = ota_oracle.cost_terms(table[0], boxes[0].numpy(), tb.numpy(), pm.numpy())
        assert np.isnan(iou).sum() == 1
        sel, gt, o_matched, M, st = ota_oracle.dynamic_k(cost, iou, flags)
        assert np.array_equal(idx[0].cpu().numpy(), sel) and np.array_equal(idx[1].cpu().numpy(), gt)
        assert np.array_equal(matched.cpu().numpy(), o_matched)


@pytest.mark.parametrize("what", ["negative width", "nan", "negative target height"])
def test_ota_device_raises_the_reference_assert_on_degenerate_boxes(what):
    """the device simOTA path used to map NaN costs to +inf and return an assignment where the reference's
    generalized_box_iou aborts the step (util/box_ops.py:76-77).  ota_cost_hip_f32 now marks such pairs, ota_dynamic_k_hip turns
    the mark into status 4 of the image, and the binding raises AssertionError behind its ONE host copy (no extra sync)."""
    g = torch.Generator().manual_seed(78)
    bs, Q, T, G = 2, 300, 16, 5
    logits = torch.randn(bs, Q, T, generator=g)
    boxes = torch.cat([torch.rand(bs, Q, 2, generator=g), 0.05 + 0.2 * torch.rand(bs, Q, 2, generator=g)], -1)
    targets = []
    for b in range(bs):
        tb = torch.cat([0.3 + 0.4 * torch.rand(G, 2, generator=g), 0.1 + 0.2 * torch.rand(G, 2, generator=g)], -1)
        pm = torch.zeros(G, T, dtype=torch.bool)
        pm[torch.arange(G), torch.arange(G)] = True
        targets.append({"boxes": tb.cuda(), "positive_map": pm.cuda()})
    m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    assert m.device_ota
    m.forward_ota({"pred_logits": logits.cuda(), "pred_boxes": boxes.cuda()}, targets)          # clean inputs: fine
    if what == "negative width":
        boxes[1, 17, 2] = -0.1
    elif what == "nan":
        boxes[0, 3, 0] = float("nan")
    else:
        targets[1]["boxes"][2, 3] = -0.05
    outputs = {"pred_logits": logits.cuda(), "pred_boxes": boxes.cuda()}
    *_, status, _sizes = m.ota_device_launch(outputs["pred_logits"].sigmoid(), outputs["pred_boxes"], targets)
    assert status.tolist() == ([4, 0] if what == "nan" else [0, 4])
    with pytest.raises(AssertionError, match="degenerate box"):
        m.forward_ota(outputs, targets)
    m.device_ota = False
    with pytest.raises(AssertionError):                               # the composition: the reference's own assert
        m.forward_ota(outputs, targets)


def _ulps(a, b):
    """Distance in float32 units in the last place (same-sign finite values)."""
    ia, ib = a.view(torch.int32).to(torch.int64), b.view(torch.int32).to(torch.int64)
    return (ia - ib).abs()


@pytest.mark.parametrize("name", matcher_names())
def test_fused_cost_matrix_is_the_compositions(name):
    """matcher_cost_hip_f32 evaluates the PyTorch composition's float32 operations in its order.  Measured against ATen on
    the MI355X (tools/matcher_cost_dbg2.py): exp, sigmoid, the L1 distance and the GIoU chain are BITWISE the composition's;
    logf of this ROCm's device library and the one PyTorch was built with differ by one unit in the last place on a third of
    the arguments, so the class term (a difference of two products with a log each) is held to 1e-6 absolute instead."""
    g, bs, outputs, targets = _case(name, device="cuda:0")
    if sum(len(t["boxes"]) for t in targets) == 0:
        pytest.skip("no targets")
    from uninext_amd import ext
    from uninext_amd.matcher import box_cxcywh_to_xyxy, focal_token_cost, generalized_box_iou
    logits = outputs["pred_logits"].flatten(0, 1)
    boxes = outputs["pred_boxes"].flatten(0, 1)
    tgt_map = torch.cat([t["positive_map"] for t in targets])
    tgt_boxes = torch.cat([t["boxes"] for t in targets])
    cls = focal_token_cost(logits.sigmoid(), tgt_map)
    l1 = torch.cdist(boxes, tgt_boxes, p=1)
    gi = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt_boxes))
    # one term at a time (the other two enter the kernel's sum as exact zeros, like here)
    assert torch.equal(ext.matcher_cost(logits, boxes, tgt_boxes, tgt_map, 0, 1, 0), (1.0 * l1 + 0.0 * cls) + 0.0 * gi)
    assert torch.equal(ext.matcher_cost(logits, boxes, tgt_boxes, tgt_map, 0, 0, 1), (0.0 * l1 + 0.0 * cls) + 1.0 * gi)
    fc = ext.matcher_cost(logits, boxes, tgt_boxes, tgt_map, 1, 0, 0)
    err = float((fc - cls).abs().max())
    print("%s: class term max |fused - composition| %.2e (max |term| %.2f)" % (name, err, float(cls.abs().max())))
    assert err < 1e-6
    full = ext.matcher_cost(logits, boxes, tgt_boxes, tgt_map, 2, 5, 2)
    assert float((full - (5 * l1 + 2 * cls + 2 * gi)).abs().max()) < 4e-6
