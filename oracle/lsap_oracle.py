"""CPU oracle for the linear sum assignment step of the Hungarian matcher.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).

The reference calls scipy.optimize.linear_sum_assignment (dd/matcher.py:257,502; SciPy is a third-party dependency that
is not vendored, `scipy>1.5.1` in the reference's setup.py:185, 1.15.3 in this image).  SciPy implements the
shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES
52(4), 2016 (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp).  This file restates that published algorithm INCLUDING
the details that decide between equal-cost optima -- the order in which the remaining columns are scanned (filled in
reverse, swap-removed), the preference for an unassigned column among ties -- so that the assignment it returns is the
one SciPy returns, index for index.  It is pinned against SciPy itself on random, integer-valued and constant matrices
(tests/test_lsap_cpu.py); the HIP kernel (include/lsap_hip.h) is then held to it and to SciPy.
"""
import numpy as np


def linear_sum_assignment(cost):
    """cost [nr, nc] (any real dtype) -> (row_ind int64 ascending, col_ind int64), len min(nr, nc)."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    if nr == 0 or nc == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    if np.isnan(cost).any() or np.isneginf(cost).any():
        raise ValueError("matrix contains invalid numeric entries")
    transpose = nc < nr
    if transpose:
        cost = np.ascontiguousarray(cost.T)
        nr, nc = nc, nr
    u, v = np.zeros(nr), np.zeros(nc)
    spc = np.empty(nc)
    path = np.full(nc, -1, np.int64)
    col4row = np.full(nr, -1, np.int64)
    row4col = np.full(nc, -1, np.int64)
    remaining = np.empty(nc, np.int64)
    for cur in range(nr):
        # ---- augmenting path from row `cur` ----
        min_val = 0.0
        num_remaining = nc
        remaining[:] = nc - 1 - np.arange(nc)          # reverse fill: a constant matrix yields the identity
        SR = np.zeros(nr, bool)
        SC = np.zeros(nc, bool)
        spc[:] = np.inf
        i, sink = cur, -1
        while sink == -1:
            index, lowest = -1, np.inf
            SR[i] = True
            for it in range(num_remaining):
                j = remaining[it]
                r = min_val + cost[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == np.inf:
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        # ---- dual update ----
        u[cur] += min_val
        for r_ in range(nr):
            if SR[r_] and r_ != cur:
                u[r_] += min_val - spc[col4row[r_]]
        for j in range(nc):
            if SC[j]:
                v[j] -= min_val - spc[j]
        # ---- augment ----
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    if transpose:
        order = np.argsort(col4row, kind="stable")
        return col4row[order].astype(np.int64), order.astype(np.int64)
    return np.arange(nr, dtype=np.int64), col4row.astype(np.int64)
