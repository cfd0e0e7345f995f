"""GPU: the matcher mirror with predictions/targets resident on the MI355X (cost terms run on the device, the
assignment on the host as in the reference) must return the same integer indices as the reference-minted fixtures."""
import numpy as np
import pytest
import torch

from golden_util import matcher_names
from test_matcher_cpu import _case
from uninext_amd.matcher import HungarianMatcherVL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", matcher_names())
def test_hungarian_on_device_matches_reference(name):
    g, bs, outputs, targets = _case(name, device="cuda:0")
    result = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2).forward(outputs, targets)
    for b, (i, j) in enumerate(result):
        assert np.array_equal(i.numpy(), g[f"hung_i_{b}"]) and np.array_equal(j.numpy(), g[f"hung_j_{b}"])


@pytest.mark.parametrize("name", [n for n in matcher_names() if "encoder" not in n])
def test_ota_on_device_matches_reference(name):
    g, bs, outputs, targets = _case(name, device="cuda:0")
    indices, matched = HungarianMatcherVL(cost_class=2, cost_bbox=5, cost_giou=2).forward_ota(outputs, targets)
    for b in range(bs):
        assert indices[b][0].is_cuda or len(targets[b]["boxes"]) == 0 or indices[b][0].device.type == "cuda"
        assert np.array_equal(indices[b][0].cpu().numpy(), g[f"ota_q_{b}"])
        assert np.array_equal(indices[b][1].cpu().numpy(), g[f"ota_g_{b}"])
        m = matched[b].cpu().numpy() if torch.is_tensor(matched[b]) else np.asarray(matched[b], dtype=np.int64)
        assert np.array_equal(m, g[f"ota_matched_{b}"])
