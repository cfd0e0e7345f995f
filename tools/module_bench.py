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

    # training: forward + backward of the layer (all parameters and the inputs require a gradient) -- the fused Function against
    # the reference's data flow (PyTorch prologue + MSDeformAttnFunction); peak memory of one step beside the time
    train = MSDeformAttn(256, 4, 8, 4).to(dev).train()
    with torch.no_grad():
        train.sampling_offsets.weight.normal_(0, 0.01)
        train.attention_weights.weight.normal_(0, 0.1)
    q_t = [q.clone().requires_grad_(True) for q in queries]
    s_t = [t.clone().requires_grad_(True) for t in srcs]
    go = torch.randn(N, S, 256, device=dev)

    def train_step():
        k = turn[0] % args.rotate
        turn[0] += 1
        out = train(q_t[k], ref, s_t[k], sh, lsi, None)
        out.backward(go)
    for fused in (True, False, True, False):
        MSDeformAttn.fuse_training_prologue = fused
        for t_ in list(train.parameters()) + q_t + s_t:
            t_.grad = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        t = timeit(train_step, max(args.reps // 3, 5))
        peak = (torch.cuda.max_memory_allocated() - base) / 2 ** 20
        print("training %-34s forward + backward %8.1f us   peak extra memory %7.1f MiB" % (
            "MSDeformAttnFusedFunction" if fused else "PyTorch prologue + Function", t, peak))
    MSDeformAttn.fuse_training_prologue = True


if __name__ == "__main__":
    main()
