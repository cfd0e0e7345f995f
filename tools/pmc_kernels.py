#!/usr/bin/env python
"""Per-kernel means of rocprofv3 PMC counters over any command (GPU box only):

    python tools/pmc_kernels.py --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES ... [--filter msda] [--tag name] -- <command ...>

One rocprofv3 pass (counters only; never combined with tracing domains, see the gpurun rules), output under
gpurun_out/pmc_<tag>; prints, per kernel name, dispatches and the mean of every counter per dispatch."""
import argparse
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    if "--" not in argv:
        raise SystemExit(__doc__)
    cut = argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", nargs="+", required=True)
    ap.add_argument("--filter", default="msda")
    ap.add_argument("--tag", default="run")
    ap.add_argument("--skip", type=int, default=0, help="drop the first N dispatches of every kernel (warm-up)")
    args = ap.parse_args(argv[:cut])
    cmd = argv[cut + 1:]
    d = os.path.join(ROOT, "gpurun_out", "pmc_" + args.tag)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    full = ["rocprofv3", "--pmc"] + args.pmc + ["-d", d, "--"] + cmd
    res = subprocess.run(full, cwd="/tmp", env=env, capture_output=True, text=True)
    sys.stdout.write(res.stdout[-3000:])
    if res.returncode != 0:
        raise SystemExit("rocprofv3 failed:\n" + res.stderr[-3000:])
    db = None
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith("_results.db"):
                p = os.path.join(root, f)
                if db is None or os.path.getmtime(p) > os.path.getmtime(db):
                    db = p
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id").fetchall()
    per = {}
    for did, name, cn, val in rows:
        e = per.setdefault(did, [name, {}])
        e[1][cn] = e[1].get(cn, 0.0) + val
    agg, seen = {}, {}
    for did in sorted(per):
        name, vals = per[did]
        if args.filter not in name:
            continue
        name = name[:name.index("(")] if "(" in name else name
        seen[name] = seen.get(name, 0) + 1
        if seen[name] <= args.skip:
            continue
        a = agg.setdefault(name, [0, {}])
        a[0] += 1
        for k, v in vals.items():
            a[1][k] = a[1].get(k, 0.0) + v
    print("# rocprofv3 --pmc %s -- %s" % (" ".join(args.pmc), " ".join(cmd)))
    for name, (n, vals) in agg.items():
        print("%s   dispatches %d" % (name, n))
        for k in args.pmc:
            if k in vals:
                print("    %-28s %16.1f" % (k, vals[k] / n))


if __name__ == "__main__":
    main()
