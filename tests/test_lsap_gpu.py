"""MI355X tests of the on-device linear sum assignment (include/lsap_hip.h): index-for-index equality with SciPy on
random, tie-heavy, constant, rectangular (both orientations), strided and batched problems, the encoder-proposal size
(22 223 x G), SciPy's error cases, and the matcher running on it against the reference-minted fixtures."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment as scipy_lsa

from golden_util import matcher_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _mats(seed, n, lo=1, hi=40):
    rng = np.random.default_rng(seed)
    for t in range(n):
        nr, nc = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        kind = t % 5
        if kind == 0:
            c = rng.standard_normal((nr, nc))
        elif kind == 1:
            c = rng.integers(0, 3, (nr, nc))
        elif kind == 2:
            c = np.round(rng.random((nr, nc)), 1)
        elif kind == 3:
            c = np.full((nr, nc), 1.5)
        else:
            c = rng.standard_normal((nr, nc))
            c[rng.random((nr, nc)) < 0.15] = np.inf
        yield c.astype(np.float32)


@pytest.mark.parametrize("seed", range(3))
def test_equals_scipy_batched(seed, dev):
    from uninext_amd import ext
    mats = list(_mats(seed, 70))            # > 2 launches of 32
    feasible = []
    for c in mats:
        try:
            feasible.append((c, scipy_lsa(c)))
        except ValueError:
            pass
    got = ext.lsap_batch([torch.from_numpy(c).to(dev) for c, _ in feasible])
    for (c, want), (r, k) in zip(feasible, got):
        assert r.dtype == torch.int64 and np.array_equal(r.cpu().numpy(), want[0]) and np.array_equal(k.cpu().numpy(), want[1]), c.shape


def test_strided_views_and_empty(dev):
    from uninext_amd import ext
    rng = np.random.default_rng(5)
    big = torch.from_numpy(rng.standard_normal((50, 37)).astype(np.float32)).to(dev)
    views = [big[:, 0:7], big[:, 7:20], big[:, 20:37], big[:, 5:5]]
    got = ext.lsap_batch(views)
    for v, (r, k) in zip(views, got):
        want = scipy_lsa(v.cpu().numpy())
        assert np.array_equal(r.cpu().numpy(), want[0]) and np.array_equal(k.cpu().numpy(), want[1])
    assert got[3][0].numel() == 0
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.lsap(big.t())


def test_encoder_proposal_size(dev):
    """22 223 proposals x 60 targets with structured costs (clusters of near-equal entries)."""
    from uninext_amd import ext
    g = torch.Generator().manual_seed(3)
    cost = torch.randn(22223, 60, generator=g)
    cost[:, 10:20] = cost[:, 10:20].round(decimals=1)              # ties inside the matrix
    cost[::7, 30] = 0.25
    want = scipy_lsa(cost.numpy())
    r, k = ext.lsap(cost.to(dev))
    assert np.array_equal(r.cpu().numpy(), want[0]) and np.array_equal(k.cpu().numpy(), want[1])
    wide = cost.t().contiguous()                                    # 60 x 22223: no transpose inside
    want = scipy_lsa(wide.numpy())
    r, k = ext.lsap(wide.to(dev))
    assert np.array_equal(r.cpu().numpy(), want[0]) and np.array_equal(k.cpu().numpy(), want[1])


def test_error_cases_follow_scipy(dev):
    from uninext_amd import ext
    with pytest.raises(ValueError, match="invalid numeric"):
        ext.lsap(torch.tensor([[1.0, float("nan")], [0.0, 1.0]], device=dev))
    with pytest.raises(ValueError, match="invalid numeric"):
        ext.lsap(torch.tensor([[1.0, float("-inf")], [0.0, 1.0]], device=dev))
    with pytest.raises(ValueError, match="infeasible"):
        ext.lsap(torch.full((3, 3), float("inf"), device=dev))
    r, k = ext.lsap(torch.tensor([[1.0, float("nan")]], device=dev), check=False)      # unchecked: no exception, no sync
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", matcher_names())
def test_matcher_on_device_lsap_matches_reference_fixture(name, dev):
    from test_matcher_cpu import _case
    from uninext_amd.matcher import HungarianMatcherVL
    g, bs, outputs, targets = _case(name, device=dev)
    m = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2)
    assert m.device_lsap
    result = m.forward(outputs, targets)
    for b, (i, j) in enumerate(result):
        assert not i.is_cuda and i.dtype == torch.int64
        assert np.array_equal(i.numpy(), g[f"hung_i_{b}"]) and np.array_equal(j.numpy(), g[f"hung_j_{b}"])
    HungarianMatcherVL.device_lsap = False          # and the host route still gives the same
    try:
        host = m.forward(outputs, targets)
    finally:
        HungarianMatcherVL.device_lsap = True
    for (i, j), (hi, hj) in zip(result, host):
        assert torch.equal(i, hi) and torch.equal(j, hj)
