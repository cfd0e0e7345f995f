"""The filing rule of the destination-side backward that waits in uninext_amd/csrc/next/ (DESIGN.md section 7), restated in
numpy (tools/proto/owner_computes_ref.py), against the C oracle: every in-image corner of every sample is applied exactly
once, by the region that owns its pixel -- for region layouts that halve per level, that do not, and that are larger than a
level.  Keeps the prototype (the reference for the device-side bins and records) from rotting."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _proto():
    spec = importlib.util.spec_from_file_location("owner_computes_ref", os.path.join(ROOT, "tools", "proto", "owner_computes_ref.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("sizes", [((16, 16), (8, 16), (4, 8), (2, 4)), ((16, 16),) * 4, ((3, 5), (8, 8), (2, 2), (64, 64))])
@pytest.mark.parametrize("flavour", ["model", "uniform"])
def test_every_corner_is_added_once_by_the_region_that_owns_it(flavour, sizes):
    assert _proto().check(((25, 42), (13, 21), (7, 11), (4, 6)), sizes, flavour, seed=11)
