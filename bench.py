#!/usr/bin/env python
"""bench.py -- throughput of the MSDeformAttn hot path of UNINEXT's R50 COCO det+seg model on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).

A "step" is one pass of the hot path over one synthetic batch of BASELINE.json configs[1]
(R50 COCO det+instance-seg inference, bs = 2, 800x1333): the 12 MultiScaleDeformableAttention
forward calls one frame pair makes -- 6 encoder self-attention calls (Lq = S = 22223) followed by 6
decoder cross-attention calls (Lq = 900) -- each on its own seeded model-like inputs,
all resident in HBM before the timed region.  value = frames/s = N_gpus * 2 * K / max-over-ranks time.
The rest of the model (backbone, projections, heads) is NOT in the step: this is the hot-path metric.

Multi-GPU: frames shard data-parallel, every rank runs the same per-rank batch (weak scaling), no
collective inside the timed region; a barrier + cuda synchronize bracket it and the time is the MAX over
ranks (all_reduce MAX over RCCL).

Extra objects on the JSON line:
  roofline      dominant kernel = the encoder forward launch.  achieved = algorithmic bytes per launch
                (N*(1024*S + 2560*Lq) B) / average launch duration measured with HIP events
                (torch.cuda.Event on the stream the kernel is launched on = torch's current stream) around
                the encoder launches INSIDE the timed region.  peak = 8 TB/s (MI355X HBM3E spec).
  cpu_baseline  the reference's CPU path (ms_deform_attn_core_pytorch, restated in oracle/msda_gridsample.py)
                timed on this box's host cores on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from uninext_amd import workloads  # noqa: E402
from uninext_amd import ext as MSDA  # noqa: E402
from uninext_amd import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ENC_LAYERS, DEC_LAYERS, BATCH = 6, 6, 2


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--flavour", default="model", choices=["model", "uniform"],
                    help="sampling-location distribution (SURVEY.md 8(d)); 'model' is the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per encoder launch from a separate rocprofv3 --pmc pass; default: the committed "
                         "profiles/traffic.json entry for the kernel that ran (see profiles/README.md)")
    return ap.parse_args()


def init_distributed(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if n_gpus > 1 and world != n_gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (n_gpus, n_gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm
    return rank, world


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(seconds, world, device="cuda"):
    """Wall time of the job = the slowest rank's time."""
    if world == 1:
        return seconds
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def build_inputs(flavour, rank):
    enc = [workloads.make_inputs("encoder", flavour, batch=BATCH, seed=100 * rank + i) for i in range(ENC_LAYERS)]
    dec = [workloads.make_inputs("decoder", flavour, batch=BATCH, seed=100 * rank + 50 + i) for i in range(DEC_LAYERS)]
    return enc, dec


def call(x):
    return MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)


def run_step(enc, dec, ev=None):
    if ev is not None:
        ev[0].record()
    for x in enc:
        call(x)
    if ev is not None:
        ev[1].record()
    for x in dec:
        call(x)


def committed_traffic(kernel):
    """PMC counters cannot be collected inside the timed run; the last committed PMC pass of the same command is
    the source (profiles/traffic.json), only used when it was taken on the kernel that ran here."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return float(rec["traffic_bytes_per_launch"]) if rec.get("kernel") == kernel else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(flavour):
    """Bounded sample of the same workload on the host: one encoder call and one decoder call of the
    reference's grid_sample path (N = 2); a step is 6 of each.  grid_sample's OpenMP scaling collapses when
    oversubscribed, so two thread counts are tried (all logical cores, and 64 when the box has more) and the
    faster one is reported together with the thread count it used."""
    from oracle.msda_gridsample import msda_gridsample
    ncpu = os.cpu_count() or 1
    inputs = {}
    for kind in ("encoder", "decoder"):
        x = workloads.make_inputs(kind, flavour, batch=BATCH, seed=7, device="cpu")
        inputs[kind] = (x, [tuple(r) for r in x["shapes"].tolist()])
    best = None
    for threads in sorted({ncpu, min(ncpu, 64)}, reverse=True):
        torch.set_num_threads(threads)
        per_step = 0.0
        with torch.no_grad():
            for kind, layers in (("encoder", ENC_LAYERS), ("decoder", DEC_LAYERS)):
                x, shapes = inputs[kind]
                msda_gridsample(x["value"], shapes, x["loc"], x["attn"])
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    msda_gridsample(x["value"], shapes, x["loc"], x["attn"])
                    ts.append(time.perf_counter() - t0)
                per_step += layers * sorted(ts)[1]
        if best is None or per_step < best[0]:
            best = (per_step, threads)
    per_step, threads = best
    return {"value": BATCH / per_step, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "median of 3 runs (after 1 warm-up) of ONE encoder call and ONE decoder call at N=2, "
                      "x6 each per step; oracle/msda_gridsample.py (= ms_deform_attn_core_pytorch), fp32, "
                      "torch.set_num_threads(%d) of %d logical cores (faster of the thread counts tried)" % (threads, ncpu)}


def main():
    args = parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU implementation)")
    rank, world = init_distributed(args.gpus)
    _lib.load()
    enc, dec = build_inputs(args.flavour, rank)
    S = enc[0]["value"].shape[1]

    call(enc[0])
    enc_kernel = _lib.last_kernel("forward")   # the kernel the encoder launches take (decoder calls may differ)
    for _ in range(max(args.warmup, 0)):
        run_step(enc, dec)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        run_step(enc, dec, events[k])
    torch.cuda.synchronize()
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world)

    if rank == 0:
        enc_ms = sum(a.elapsed_time(b) for a, b in events) / (args.steps * ENC_LAYERS)  # per encoder launch
        alg_bytes = workloads.algorithmic_bytes_forward(BATCH, S, S)
        achieved = alg_bytes / (enc_ms * 1e-3) / 1e9
        out = {
            "metric": "frames/sec COCO det+seg R50 1333x800 (MSDeformAttn hot path); MSDeformAttn HBM GB/s",
            "value": world * BATCH * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: R50 COCO det+seg inference bs=2 800x1333 -- hot path only: "
                            "6 encoder (Lq=S=22223) + 6 decoder (Lq=900) MSDeformAttn forward calls per step, "
                            "M=8 D=32 L=4 P=4, '%s' sampling locations" % args.flavour,
                "frames_per_step_per_gpu": BATCH,
                "parallelism": "dp%d (independent replicas, no collective in the timed region)" % world,
                "kernel": enc_kernel,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": args.traffic_bytes if args.traffic_bytes is not None else committed_traffic(enc_kernel),
                "kernel": enc_kernel, "launch_us": 1e3 * enc_ms, "algorithmic_bytes": alg_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.flavour)
        print(json.dumps(out), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
