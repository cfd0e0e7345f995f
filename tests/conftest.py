import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def split_bf16_paths():
    """The split-bf16 inference paths (MSDeformAttn.fast_linear, MaskHeadSmallConv / PatchEmbed exact_fp32 = False) are
    OPT-IN since round 4 -- the modules default to the reference's fp32 arithmetic.  Test files about the fast kernels
    switch them on for their tests (pytestmark = pytest.mark.usefixtures("split_bf16_paths"))."""
    from uninext_amd.backbone import PatchEmbed
    from uninext_amd.mask_head import MaskHeadSmallConv
    from uninext_amd.modules import MSDeformAttn
    old = (MSDeformAttn.fast_linear, MaskHeadSmallConv.exact_fp32, PatchEmbed.exact_fp32)
    MSDeformAttn.fast_linear, MaskHeadSmallConv.exact_fp32, PatchEmbed.exact_fp32 = True, False, False
    try:
        yield
    finally:
        MSDeformAttn.fast_linear, MaskHeadSmallConv.exact_fp32, PatchEmbed.exact_fp32 = old
