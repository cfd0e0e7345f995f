"""Host logic behind the packed-weight / shape-check caches (uninext_amd/_cache.py) and the autograd routing of
MSDeformAttn (round-1 advisor findings): inference tensors carry no version counter, recycled addresses must not
skip the shape assert, `.data` edits are covered by explicit invalidation, and a layer with ANY trainable parameter
must not take the forward-only fused entry points."""
import torch

from uninext_amd import _cache
from uninext_amd.modules import MSDeformAttn


def test_tensor_version_inference_tensor():
    with torch.inference_mode():
        t = torch.ones(3)
    assert t.is_inference() and _cache.tensor_version(t) == -1
    u = torch.ones(3)
    v0 = _cache.tensor_version(u)
    u.add_(1)
    assert _cache.tensor_version(u) == v0 + 1


def test_packed_weight_cache_rebuilds_and_invalidates():
    lin = torch.nn.Linear(4, 4)
    calls = []
    pack = lambda w: calls.append(1) or w.clone()
    a = _cache.packed_weight(lin, pack)
    assert _cache.packed_weight(lin, pack) is a and len(calls) == 1
    with torch.no_grad():
        lin.weight.mul_(2.0)                       # bumps the version
    b = _cache.packed_weight(lin, pack)
    assert b is not a and len(calls) == 2
    lin.weight.data.mul_(2.0)                      # invisible to the version counter ...
    assert _cache.packed_weight(lin, pack) is b
    _cache.invalidate_packed(lin)                  # ... hence the explicit hook
    assert _cache.packed_weight(lin, pack) is not b and len(calls) == 3


def test_packed_weight_on_inference_parameters():
    with torch.inference_mode():
        lin = torch.nn.Linear(4, 4)
    assert lin.weight.is_inference()
    w = _cache.packed_weight(lin, lambda t: t.clone())     # must not touch ._version
    assert _cache.packed_weight(lin, lambda t: t.clone()) is w


def test_owner_modules_drop_caches_on_mode_switch_and_load():
    m = MSDeformAttn(64, 4, 2, 4)
    m.value_proj.__dict__["_msda_packed"] = ("key", "stale")
    m.eval()
    assert "_msda_packed" not in m.value_proj.__dict__
    m.value_proj.__dict__["_msda_packed"] = ("key", "stale")
    m.load_state_dict(m.state_dict())
    assert "_msda_packed" not in m.value_proj.__dict__


def test_checked_once_is_per_tensor_object():
    c = _cache.CheckedOnce(limit=4)
    t = torch.tensor([[2, 3]])
    assert not c.hit(t, 6)
    c.add(t, 6)
    assert c.hit(t, 6) and not c.hit(t, 7)
    t.add_(1)                                       # in-place edit: checked again
    assert not c.hit(t, 6)
    u = torch.tensor([[2, 3]])                      # an equal but different tensor object is not a hit
    assert not c.hit(u, 6)
    key = id(t)
    del t                                           # a dead entry can never validate a new tensor that reuses the id
    e = c._seen.get(key)
    assert e is None or e[0]() is None
    with torch.inference_mode():
        w = torch.tensor([[2, 3]])
    c.add(w, 6)
    assert c.hit(w, 6)
    for i in range(10):                             # bounded
        c.add(torch.tensor([[i, 1]]), i)
    assert len(c._seen) <= 6


def test_shape_assert_still_fires_on_cpu():
    import pytest
    with pytest.raises(AssertionError):
        MSDeformAttn._check_shapes(torch.tensor([[2, 3], [1, 1]]), 8)
    MSDeformAttn._check_shapes(torch.tensor([[2, 3], [1, 1]]), 7)


def test_partial_finetuning_never_takes_forward_only_paths():
    m = MSDeformAttn(64, 4, 2, 4)
    x = torch.zeros(1, 5, 64)
    for p in m.parameters():
        p.requires_grad_(False)
    assert not m._records_grad(x, None)                       # fully frozen, inputs without grad: inference paths allowed
    m.sampling_offsets.weight.requires_grad_(True)            # only the offsets are fine-tuned; value_proj frozen
    assert m._records_grad(x, None)
    with torch.no_grad():
        assert not m._records_grad(x, None)
    m.sampling_offsets.weight.requires_grad_(False)
    assert m._records_grad(x.clone().requires_grad_(True))
