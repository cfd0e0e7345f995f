#!/usr/bin/env python
"""DeformableTransformerEncoderLayer at the R50 COCO shapes (bs 2, 22 223 tokens x 256, d_ffn 1024), inference:
this repo's layer vs the same layer computed with PyTorch-ROCm ops around the operator (the reference's data flow).

    python tools/encoder_layer_bench.py [--reps 20] [--rotate 4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uninext_amd import workloads  # noqa: E402
from uninext_amd.modules import DeformableTransformerEncoderLayer, MSDeformAttn  # noqa: E402
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
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rotate", type=int, default=4)
    args = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    levels = workloads.R50_LEVELS_INFER
    S = sum(h * w for h, w in levels)
    N = 2
    layer = DeformableTransformerEncoderLayer().to(dev).eval()
    with torch.no_grad():
        layer.self_attn.sampling_offsets.weight.normal_(0, 0.01)
        layer.self_attn.attention_weights.weight.normal_(0, 0.1)
    srcs = [torch.randn(N, S, 256, device=dev) for _ in range(args.rotate)]
    pos = torch.randn(N, S, 256, device=dev) * 0.3
    ref = workloads.encoder_reference_points(levels, dev)[None, :, None, :].expand(N, S, 4, 2).contiguous()
    sh, lsi = workloads.level_tensors(levels, dev)
    turn = [0]

    def ours():
        k = turn[0] % args.rotate
        turn[0] += 1
        return layer(srcs[k], pos, ref, sh, lsi, None)

    def torch_ops():
        k = turn[0] % args.rotate
        turn[0] += 1
        src = srcs[k]
        src2 = layer.self_attn(src + pos, ref, src, sh, lsi, None)
        src = layer.norm1(src + src2)
        src2 = layer.linear2(torch.relu(layer.linear1(src)))
        return layer.norm2(src + src2)

    with torch.no_grad():
        t_fast = timeit(ours, args.reps)
        MSDeformAttn.fast_linear = False
        t_lib = timeit(torch_ops, args.reps)                 # library GEMMs, fused sampling kernel, PyTorch add / LayerNorm
        MSDeformAttn.fuse_prologue = False
        t_ref = timeit(torch_ops, args.reps)                 # + PyTorch prologue: the reference's data flow
        MSDeformAttn.fast_linear = True
        MSDeformAttn.fuse_prologue = True
        a, b = ours(), None
        turn[0] -= 1
        MSDeformAttn.fast_linear = False
        b = torch_ops()
        MSDeformAttn.fast_linear = True
        err = float((a - b).abs().max()) / float(b.abs().max())
        # the same layer replayed from a HIP graph (static shapes: every kernel of the path only enqueues work)
        static_src = srcs[0].clone()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                layer(static_src, pos, ref, sh, lsi, None)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            static_out = layer(static_src, pos, ref, sh, lsi, None)

        def replay():
            k = turn[0] % args.rotate
            turn[0] += 1
            static_src.copy_(srcs[k])
            graph.replay()
        t_graph = timeit(replay, args.reps)
        static_src.copy_(srcs[0]); graph.replay(); torch.cuda.synchronize()
        gerr = float((static_out - layer(srcs[0], pos, ref, sh, lsi, None)).abs().max())
    print("HIP-graph replay of this repo's layer (incl. the 45 MB input copy): %.1f us, max diff vs eager %.1e" % (t_graph, gerr))
    print("encoder layer (bs 2, %d tokens): this repo %.1f us | PyTorch ops + fused sampling %.1f us | reference data flow "
          "(PyTorch ops + operator) %.1f us | rel diff %.1e" % (S, t_fast, t_lib, t_ref, err))
    print("six layers (extrapolated): %.2f ms vs %.2f ms" % (6e-3 * t_fast, 6e-3 * t_ref))
    # six DISTINCT layers back to back, as DeformableTransformerEncoder.forward runs them (deformable_transformer_dino.py)
    layers = [layer] + [DeformableTransformerEncoderLayer().to(dev).eval() for _ in range(5)]
    with torch.no_grad():
        for l in layers[1:]:
            l.self_attn.sampling_offsets.weight.normal_(0, 0.01)
            l.self_attn.attention_weights.weight.normal_(0, 0.1)

        def encoder(fast):
            k = turn[0] % args.rotate
            turn[0] += 1
            out = srcs[k]
            for l in layers:
                if fast:
                    out = l(out, pos, ref, sh, lsi, None)
                else:
                    src2 = l.self_attn(out + pos, ref, out, sh, lsi, None)
                    out = l.norm1(out + src2)
                    out = l.norm2(out + l.linear2(torch.relu(l.linear1(out))))
            return out
        t6 = timeit(lambda: encoder(True), max(args.reps // 2, 5))
        MSDeformAttn.fast_linear = False
        MSDeformAttn.fuse_prologue = False
        t6_ref = timeit(lambda: encoder(False), max(args.reps // 2, 5))
        MSDeformAttn.fast_linear = True
        MSDeformAttn.fuse_prologue = True
    print("six distinct layers back to back (measured): %.2f ms vs %.2f ms (PyTorch ops + operator)" % (t6 * 1e-3, t6_ref * 1e-3))


if __name__ == "__main__":
    main()
