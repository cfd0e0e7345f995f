#!/usr/bin/env python
"""Assignment step of HungarianMatcherVL.forward at the sizes of the R50 configs (GPU box only): include/lsap_hip.h on
the device vs the reference's `C.cpu()` + scipy.optimize.linear_sum_assignment per image.

    python tools/lsap_bench.py
"""
import os
import sys
import time

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uninext_amd import ext  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, Q, Gs in (("decoder queries, bs 2", 900, (12, 35)), ("decoder queries, bs 2, crowded", 900, (80, 100)),
                        ("encoder proposals, bs 2", 22223, (12, 35)), ("encoder proposals, bs 2, crowded", 22223, (80, 100))):
        total = sum(Gs)
        cost = torch.randn(len(Gs), Q, total, generator=g).to(dev)
        blocks = [blk[b] for b, blk in enumerate(cost.split(list(Gs), -1))]
        ext.lsap_batch(blocks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            got = ext.lsap_batch(blocks, check=False)
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(3):
            c = cost.cpu()
            want = [linear_sum_assignment(blk[b]) for b, blk in enumerate(c.split(list(Gs), -1))]
        t_host = (time.perf_counter() - t0) / 3
        same = all(np.array_equal(a.cpu().numpy(), w[0]) and np.array_equal(b.cpu().numpy(), w[1]) for (a, b), w in zip(got, want))
        print("%-36s Q=%5d G=%-9s device %8.2f ms | C.cpu() + SciPy %8.2f ms | identical indices: %s"
              % (name, Q, Gs, t_dev * 1e3, t_host * 1e3, same))


if __name__ == "__main__":
    main()
