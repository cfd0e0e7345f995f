"""Per-module caches of packed weights (include/linear_hip.h, conv3x3_hip.h, patch_embed_hip.h).

A packed copy is keyed on the parameter's storage address, shape, device and autograd version counter.  Two things
the version counter does not see are handled explicitly:
  * inference tensors (parameters created or loaded under torch.inference_mode()) do not track a version at all --
    reading `_version` raises -- so they are keyed without it;
  * `param.data.copy_()` / `param.data.mul_()` edit the storage behind autograd's back.  Owners of cached modules
    therefore drop the caches on `train()` / `eval()` and after `load_state_dict`, and `invalidate_packed(module)`
    is public for callers that edit `.data` by hand.
"""
import weakref

import torch


def tensor_version(t):
    """`t._version`, or -1 for inference tensors (no version counter; `_version` raises on them)."""
    return -1 if t.is_inference() else t._version


def packed_weight(module, pack_fn, slot="_msda_packed"):
    """Packed copy of module.weight produced by pack_fn(weight), cached on the module.  `slot`: the cache entry -- two
    packings of one weight (split-bf16 and exact fp32) keep one each ("_msda_packed", "_msda_packed_exact")."""
    w = module.weight
    key = (w.data_ptr(), tensor_version(w), str(w.device), tuple(w.shape))
    cache = module.__dict__.get(slot)
    if cache is None or cache[0] != key:
        cache = (key, pack_fn(w.detach().contiguous()))
        module.__dict__[slot] = cache
    return cache[1]


def packed_weight_pair(owner, lin_a, lin_b, pack_fn):
    """(packed copy of the row-wise concatenation [lin_a.weight; lin_b.weight], concatenated biases or None), cached on
    `owner` and rebuilt when either parameter changes -- two Linear layers on the same input run as one product."""
    ws = (lin_a.weight, lin_b.weight, lin_a.bias, lin_b.bias)
    key = tuple((t.data_ptr(), tensor_version(t), str(t.device), tuple(t.shape)) if t is not None else None for t in ws)
    cache = owner.__dict__.get("_msda_packed_pair")
    if cache is None or cache[0] != key:
        w = torch.cat([lin_a.weight.detach(), lin_b.weight.detach()], 0).contiguous()
        if lin_a.bias is not None and lin_b.bias is not None:
            b = torch.cat([lin_a.bias.detach(), lin_b.bias.detach()], 0).contiguous()
        elif lin_a.bias is None and lin_b.bias is None:
            b = None
        else:
            zeros = lambda lin: torch.zeros(lin.weight.shape[0], dtype=lin.weight.dtype, device=lin.weight.device)
            b = torch.cat([lin_a.bias.detach() if lin_a.bias is not None else zeros(lin_a),
                           lin_b.bias.detach() if lin_b.bias is not None else zeros(lin_b)], 0).contiguous()
        cache = (key, (pack_fn(w), b))
        owner.__dict__["_msda_packed_pair"] = cache
    return cache[1]


def invalidate_packed(root):
    """Drop every packed-weight cache under `root` (an nn.Module); the next inference call re-packs."""
    for m in root.modules():
        m.__dict__.pop("_msda_packed", None)
        m.__dict__.pop("_msda_packed_exact", None)
        m.__dict__.pop("_msda_packed_pair", None)


class CachedModuleMixin:
    """nn.Module mix-in: caches under this module die on train()/eval() and on load_state_dict."""

    def train(self, mode=True):
        invalidate_packed(self)
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        invalidate_packed(self)
        return super()._load_from_state_dict(*args, **kwargs)


class CheckedOnce:
    """Remembers which (tensor object, version, extra) combinations passed a host-side check, without keeping the
    tensors alive and without trusting recycled addresses: an entry is valid only while its weak reference still
    points at the very same tensor object."""

    def __init__(self, limit=64):
        self._seen = {}
        self._limit = limit

    def hit(self, t, extra):
        e = self._seen.get(id(t))
        return e is not None and e[0]() is t and e[1] == (tensor_version(t), extra)

    def add(self, t, extra):
        if len(self._seen) > self._limit:
            self._seen = {k: v for k, v in self._seen.items() if v[0]() is not None}
            if len(self._seen) > self._limit:
                self._seen.clear()
        self._seen[id(t)] = (weakref.ref(t), (tensor_version(t), extra))
