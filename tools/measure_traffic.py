#!/usr/bin/env python
"""HBM traffic per launch of the contract kernels, by the recipe of MI355X_MICROARCH.md (HBM section):
separate `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes (the TCC slots do not fit both), FETCH_SIZE
doubled (gfx950 tallies the 128-byte read requests of wide loads as 64 B), per launch.

    python tools/measure_traffic.py [--out profiles/traffic.json] [--sq profiles/rNN_sq_pmc.txt] [--reps 6]

Runs on the GPU box (from the repository root; rocprofv3 output goes to gpurun_out/traffic_prof).  The parent
starts `rocprofv3 ... -- python tools/measure_traffic.py --child` once per counter set; the child launches, in this
order and separated by a marker kernel, `reps` encoder forward calls (variant 0: the window kernel on these inputs), `reps` with the gather
kernel pinned, `reps` encoder backward calls without a call context (msda_bwd_tiled), `reps` with the context of a call site whose forward calls
reported near samples (msda_bwd_win) and `reps` decoder backward calls (forward: BASELINE configs[1] shapes; backward: configs[4] training
shapes; model-like locations, rotating input sets).
The parent splits the dispatch-ordered counter rows into one run of msda:: kernels per call and writes

    {"source_hash": <bench.kernel_source_hash()>, "git": <HEAD>, "entries": {"forward_encoder": {"kernel": ...,
      "fetch_size_kib": ..., "write_size_kib": ..., "traffic_bytes_per_launch": (2*FETCH + WRITE)*1024, ...}, ...}}

bench.py quotes an entry as roofline.traffic only when both the kernel name and the source hash match the run.
"""
import argparse
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ("forward_encoder", "forward_encoder_lg3", "backward_encoder", "backward_encoder_win", "backward_decoder")
SQ_COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
               "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_LDS_BANK_CONFLICT"]


def child(reps):
    import torch
    import bench
    from uninext_amd import _lib, workloads
    _lib.load()
    marker = torch.zeros(64, device="cuda")
    enc = [workloads.make_inputs("encoder", "model", batch=2, seed=300 + i) for i in range(3)]
    # the backward phases run on the TRAINING shapes (800 x 1344: S = 22323, decoder Lq = 1100), like bench.py's backward legs
    enc_t = [workloads.make_workload("r50_train_encoder", "model", seed=320 + i) for i in range(3)]
    dec_t = [workloads.make_workload("r50_train_decoder", "model", seed=350 + i) for i in range(3)]
    names = {}
    from uninext_amd import ext
    for x in enc_t:                     # the one-off geometry check of a shapes tensor launches torch kernels: not between two msda
        ext._geometry_checked(x["shapes"], x["lsi"], x["value"].shape[1])   # launches of one marker-delimited call
    # one untimed call of each kind first (dynamic-LDS opt-in, lazy module load) -- also separated by markers
    for phase in range(len(PHASES)):
        xs = enc if phase < 2 else enc_t if phase < 4 else dec_t
        bufs = []
        for i, x in enumerate(xs):
            g = torch.Generator().manual_seed(400 + i)
            go = torch.randn(x["value"].shape[0], x["loc"].shape[1], 256, generator=g).cuda()
            bufs.append((go, torch.zeros_like(x["value"]), torch.empty_like(x["loc"]), torch.empty_like(x["attn"])))
        for r in range(reps + 1):       # call 0 of every phase is the warm-up; the parent drops it
            x, b = xs[r % 3], bufs[r % 3]
            b[1].zero_()
            marker.add_(1.0)
            if phase < 2:
                _lib.set_variant("forward", "msda_fwd_lg3" if phase == 1 else "auto")
                bench.call(x)
                _lib.set_variant("forward", "auto")
            else:
                if phase == 3 and r == 0:   # the forward calls of call site 1 (inside the dropped warm-up call's run of
                    for xx in xs:           # kernels): what lets this phase's backward calls take msda_bwd_win
                        for _ in range(4):
                            bench.call(xx, 1)
                bench.backward_call(x, *b, site=1 if phase == 3 else -1)
            marker.add_(1.0)
        names[PHASES[phase]] = _lib.last_kernel("forward" if phase < 2 else "backward")
        del bufs
    torch.cuda.synchronize()
    print("CHILD_KERNELS " + json.dumps(names), flush=True)


def find_db(d):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith("_results.db"):
                return os.path.join(root, f)
    raise SystemExit("no rocprofv3 results database under " + d)


def runs_of_calls(db):
    """[{counter: summed value over the kernels of the call, '_names': [...]}] in dispatch order."""
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection "
                     "order by dispatch_id").fetchall()
    per_dispatch, order = {}, []
    for did, name, cn, val in rows:
        if did not in per_dispatch:
            per_dispatch[did] = (name, {})
            order.append(did)
        per_dispatch[did][1][cn] = per_dispatch[did][1].get(cn, 0.0) + val
    calls, cur = [], None
    for did in order:
        name, vals = per_dispatch[did]
        if "msda::" in name:
            if cur is None:
                cur = {"_names": []}
                calls.append(cur)
            cur["_names"].append(name[:name.index("(")] if "(" in name else name)
            for k, v in vals.items():
                cur[k] = cur.get(k, 0.0) + v
        else:
            cur = None
    return calls


def profile(pmc, reps, outdir, tag):
    d = os.path.join(outdir, tag)
    cmd = ["rocprofv3", "--pmc"] + pmc + ["-d", d, "--", sys.executable, os.path.abspath(__file__), "--child",
                                           "--reps", str(reps)]
    env = dict(os.environ, TMPDIR="/tmp")
    res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if res.returncode != 0:
        raise SystemExit("rocprofv3 failed:\n" + res.stdout[-2000:] + res.stderr[-2000:])
    names = {}
    for line in res.stdout.splitlines():
        if line.startswith("CHILD_KERNELS "):
            names = json.loads(line[len("CHILD_KERNELS "):])
    calls = runs_of_calls(find_db(d))
    if len(calls) != len(PHASES) * (reps + 1):
        raise SystemExit("expected %d msda calls in the trace, found %d" % (len(PHASES) * (reps + 1), len(calls)))
    per_phase = {}
    for p, phase in enumerate(PHASES):
        sel = calls[p * (reps + 1) + 1:(p + 1) * (reps + 1)]      # drop the warm-up call
        per_phase[phase] = {k: sum(c.get(k, 0.0) for c in sel) / len(sel) for k in pmc}
        per_phase[phase]["_names"] = sel[0]["_names"]
    return names, per_phase


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--sq", default="", help="also collect SQ counters and write this text summary")
    ap.add_argument("--workdir", default=os.path.join(ROOT, "gpurun_out", "traffic_prof"))
    args = ap.parse_args()
    if args.child:
        return child(args.reps)
    import bench
    from uninext_amd import workloads
    os.makedirs(args.workdir, exist_ok=True)
    names, fetch = profile(["FETCH_SIZE"], args.reps, args.workdir, "fetch")
    _, write = profile(["WRITE_SIZE"], args.reps, args.workdir, "write")
    S = sum(h * w for h, w in workloads.R50_LEVELS_INFER)
    St = sum(h * w for h, w in workloads.R50_LEVELS_TRAIN)
    alg = {"forward_encoder": workloads.algorithmic_bytes_forward(2, S, S),
           "forward_encoder_lg3": workloads.algorithmic_bytes_forward(2, S, S),
           "backward_encoder": workloads.algorithmic_bytes_backward(2, St, St),
           "backward_encoder_win": workloads.algorithmic_bytes_backward(2, St, St),
           "backward_decoder": workloads.algorithmic_bytes_backward(2, St, 1100)}
    try:
        git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        git = ""
    rec = {"source_hash": bench.kernel_source_hash(), "git": git or "(snapshot without .git)",
           "recipe": "tools/measure_traffic.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, KiB per dispatch "
                     "summed over the kernels of one call, mean of %d calls; traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                     "(gfx950: FETCH_SIZE tallies 128-B read requests as 64 B)" % args.reps,
           "entries": {}}
    for phase in PHASES:
        f, w = fetch[phase]["FETCH_SIZE"], write[phase]["WRITE_SIZE"]
        t = (2.0 * f + w) * 1024.0
        rec["entries"][phase] = {"kernel": names.get(phase, ""), "kernels_in_call": fetch[phase]["_names"],
                                 "fetch_size_kib": f, "write_size_kib": w, "fetch_correction": 2.0,
                                 "traffic_bytes_per_launch": t, "algorithmic_bytes": alg[phase],
                                 "traffic_over_algorithmic": t / alg[phase]}
    with open(args.out, "w") as fh:
        json.dump(rec, fh, indent=1)
        fh.write("\n")
    print(json.dumps(rec, indent=1))
    if args.sq:
        _, sq = profile(SQ_COUNTERS, args.reps, args.workdir, "sq")
        with open(args.sq, "w") as fh:
            fh.write("# tools/measure_traffic.py --sq: rocprofv3 --pmc %s (one pass), mean per call over %d calls, source hash %s\n"
                     % (" ".join(SQ_COUNTERS), args.reps, rec["source_hash"]))
            for phase in PHASES:
                fh.write("%s  %s\n" % (phase, ",".join(sq[phase]["_names"])))
                for k in SQ_COUNTERS:
                    fh.write("    %-24s %16.1f\n" % (k, sq[phase][k]))
                wc = sq[phase]["SQ_WAVE_CYCLES"]
                if wc > 0:
                    fh.write("    -> waiting (s_waitcnt / barrier) %.1f %% of wave cycles, issue-stalled %.1f %%, LDS bank conflicts %.1f %% of LDS-active\n"
                             % (100 * sq[phase]["SQ_WAIT_ANY"] / wc, 100 * sq[phase]["SQ_WAIT_INST_ANY"] / wc,
                                100 * sq[phase]["SQ_LDS_BANK_CONFLICT"] / max(1.0, sq[phase]["SQ_ACTIVE_INST_LDS"])))
        print(open(args.sq).read())


if __name__ == "__main__":
    main()
