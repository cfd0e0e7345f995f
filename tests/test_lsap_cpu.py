"""CPU tests of the assignment step: the restated algorithm (oracle/lsap_oracle.py) against SciPy itself -- this pins
the oracle -- and the C ABI surface of include/lsap_hip.h (no GPU work)."""
import os
import re

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment as scipy_lsa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cases(seed, n):
    rng = np.random.default_rng(seed)
    for t in range(n):
        nr, nc = int(rng.integers(1, 30)), int(rng.integers(1, 30))
        kind = t % 5
        if kind == 0:
            yield rng.standard_normal((nr, nc))
        elif kind == 1:
            yield rng.integers(0, 3, (nr, nc)).astype(np.float64)        # heavy ties
        elif kind == 2:
            yield np.round(rng.random((nr, nc)), 1)
        elif kind == 3:
            yield np.full((nr, nc), 1.5)                                  # constant: SciPy returns the identity
        else:
            c = rng.standard_normal((nr, nc)).astype(np.float32)          # what the matcher produces
            c[rng.random((nr, nc)) < 0.2] = np.inf                        # forbidden pairs
            yield c


@pytest.mark.parametrize("seed", range(4))
def test_oracle_is_scipy_index_for_index(seed):
    from oracle import lsap_oracle
    for c in cases(seed, 250):
        try:
            want = scipy_lsa(c)
        except ValueError as e:
            with pytest.raises(ValueError, match=str(e).split()[0]):
                lsap_oracle.linear_sum_assignment(c)
            continue
        got = lsap_oracle.linear_sum_assignment(c)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), c.shape


def test_oracle_rejects_what_scipy_rejects():
    from oracle import lsap_oracle
    for bad in (np.array([[1.0, np.nan]]), np.array([[1.0, -np.inf]])):
        with pytest.raises(ValueError, match="invalid numeric"):
            scipy_lsa(bad)
        with pytest.raises(ValueError, match="invalid numeric"):
            lsap_oracle.linear_sum_assignment(bad)
    with pytest.raises(ValueError, match="infeasible"):
        lsap_oracle.linear_sum_assignment(np.full((2, 2), np.inf))
    assert lsap_oracle.linear_sum_assignment(np.zeros((0, 3)))[0].size == 0


def test_header_symbols_and_argument_errors():
    from uninext_amd import _lib
    text = open(os.path.join(ROOT, "include", "lsap_hip.h")).read()
    assert set(re.findall(r"\b(lsap_hip_\w+)\s*\(", text)) == set(_lib.LSAP_EXPORTS)
    assert int(re.search(r"#define LSAP_HIP_MAX_BATCH (\d+)", text).group(1)) == _lib.LSAP_MAX_BATCH
    lib = _lib.load()
    for sym in _lib.LSAP_EXPORTS:
        assert hasattr(lib, sym)
    assert lib.lsap_hip_workspace_bytes(100, 22223) == lib.lsap_hip_workspace_bytes(22223, 100) > 100 * 22223 * 8
    one = 16
    assert lib.lsap_hip_f32(one, 2, 4, 4, one, one, one, one, None) == -2          # ld < cols
    assert lib.lsap_hip_f32(None, 4, 4, 4, one, one, one, one, None) == -1
    assert lib.lsap_hip_batch_f32(33, None, None, None, None, None, None, None, None, None) == -2


def test_matcher_cost_header_symbols_and_argument_errors():
    """include/matcher_cost_hip.h: the symbol is exported and the argument checks answer without a GPU."""
    from uninext_amd import _lib
    text = open(os.path.join(ROOT, "include", "matcher_cost_hip.h")).read()
    assert set(re.findall(r"\b(matcher_cost_hip_\w+)\s*\(", text)) == set(_lib.MATCHER_COST_EXPORTS)
    lib = _lib.load()
    for sym in _lib.MATCHER_COST_EXPORTS:
        assert hasattr(lib, sym)
    one = 16
    assert lib.matcher_cost_hip_f32(one, one, one, one, one, -1, 4, 4, 1.0, 1.0, 1.0, one, None) == -2
    assert lib.matcher_cost_hip_f32(None, one, one, one, one, 4, 4, 4, 1.0, 1.0, 1.0, one, None) == -1
    assert lib.matcher_cost_hip_f32(None, None, None, None, None, 0, 4, 4, 1.0, 1.0, 1.0, None, None) == 0     # nothing to do
    assert b"matcher_cost" in lib.msda_hip_last_error()


def test_ota_header_symbols_and_argument_errors():
    """include/ota_hip.h: both symbols are exported and the argument checks answer without a GPU."""
    import ctypes
    from uninext_amd import _lib
    text = open(os.path.join(ROOT, "include", "ota_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert set(re.findall(r"\b(ota_\w+_hip\w*)\s*\(", text)) == set(_lib.OTA_EXPORTS)
    assert int(re.search(r"#define OTA_HIP_MAX_BATCH (\d+)", text).group(1)) == _lib.OTA_MAX_BATCH
    lib = _lib.load()
    for sym in _lib.OTA_EXPORTS:
        assert hasattr(lib, sym)
    one = 16
    off = (ctypes.c_int32 * 3)(0, 2, 5)
    assert lib.ota_cost_hip_f32(one, one, one, one, off, 65, 4, 4, one, one, one, None) == -2            # batch > OTA_HIP_MAX_BATCH
    assert lib.ota_cost_hip_f32(one, one, one, one, None, 2, 4, 4, one, one, one, None) == -1            # no offsets
    assert lib.ota_cost_hip_f32(None, one, one, one, off, 2, 4, 4, one, one, one, None) == -1
    assert lib.ota_cost_hip_f32(one, one, one, one, (ctypes.c_int32 * 3)(0, 3, 2), 2, 4, 4, one, one, one, None) == -2   # decreasing
    assert lib.ota_cost_hip_f32(one, one, one, one, (ctypes.c_int32 * 3)(1, 3, 4), 2, 4, 4, one, one, one, None) == -2   # gt_off[0] != 0
    assert lib.ota_cost_hip_f32(None, None, None, None, (ctypes.c_int32 * 3)(0, 0, 0), 2, 4, 4, None, None, None, None) == 0   # no targets
    assert lib.ota_dynamic_k_hip(one, one, one, one, (ctypes.c_int32 * 2)(0, 5000), 1, 4, 10, one, one, one, one, one, None) == -2   # > 4096 targets
    assert lib.ota_dynamic_k_hip(one, one, one, one, off, 2, 4, 10, one, one, one, None, one, None) == -1
    assert b"ota" in lib.msda_hip_last_error()
