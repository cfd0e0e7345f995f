#!/usr/bin/env python
"""MSDeformAttn module-level timing on the GPU box: fused prologue vs the two-step path (inference, no_grad).

    python tools/module_bench.py [--reps 30]

Encoder call of the R50 COCO config: N = 2, S = Lq = 22223, d_model 256.  Reports the whole layer
(4 Linear GEMMs + prologue + sampling) and the part the fusion touches (everything between the Linear outputs and
the sampled tensor)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uninext_amd import workloads  # noqa: E402
from uninext_amd.modules import MSDeformAttn  # noqa: E402
MSDeformAttn.fast_linear = True   # opt-in since round 4: this tool times the split-bf16 projections unless it says otherwise


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many distinct (query, src) sets (cold caches)")
    args = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    levels = workloads.R50_LEVELS_INFER
    S = sum(h * w for h, w in levels)
    N = 2
    layer = MSDeformAttn(256, 4, 8, 4).to(dev).eval()
    with torch.no_grad():
        layer.sampling_offsets.weight.normal_(0, 0.01)
        layer.attention_weights.weight.normal_(0, 0.1)
    srcs = [torch.randn(N, S, 256, device=dev) for _ in range(args.rotate)]
    queries = [t + torch.randn(N, S, 256, device=dev) * 0.1 for t in srcs]
    src, query = srcs[0], queries[0]
    turn = [0]

    def whole_layer():
        k = turn[0] % args.rotate
        turn[0] += 1
        return layer(queries[k], ref, srcs[k], sh, lsi, None)
    ref = workloads.encoder_reference_points(levels, dev)[None, :, None, :].expand(N, S, 4, 2).contiguous()
    sh, lsi = workloads.level_tensors(levels, dev)
    with torch.no_grad():
        for fuse in (True, False):
            MSDeformAttn.fuse_prologue = fuse
            whole = timeit(whole_layer, args.reps)
            value = layer.value_proj(src).view(N, S, 8, 32)
            off, lg = layer.sampling_offsets(query), layer.attention_weights(query)
            if fuse:
                from uninext_amd import ext
                part = timeit(lambda: ext.ms_deform_attn_forward_fused(value, sh, lsi, ref, off, lg, 4), args.reps)
            else:
                part = timeit(lambda: layer._sample_autograd(value, sh, lsi, ref, off, lg), args.reps)
            print("%-18s whole layer %8.1f us   prologue+sampling %8.1f us" % ("fused" if fuse else "two-step", whole, part))
    MSDeformAttn.fuse_prologue = True


if __name__ == "__main__":
    main()
