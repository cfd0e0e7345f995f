#!/usr/bin/env python
"""Summarise rocprofv3 result databases (ROCm 7.2 writes SQLite `*_results.db`) into small text files for
profiles/.  Usage:
    python tools/rocprof_summary.py trace <results.db>          # per (kernel, grid) call count / avg / min / max us
    python tools/rocprof_summary.py pmc <results.db> [...]      # per (kernel, grid, counter) average value
"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name else name


def trace(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, sgpr_count, count(*), "
                     "avg(duration), min(duration), max(duration), sum(duration) from kernels "
                     "group by name, grid_x, grid_y order by sum(duration) desc").fetchall()
    total = sum(r[-1] for r in rows)
    print("%-58s %-14s %5s %6s %5s %6s %10s %10s %10s %6s" % ("kernel", "grid(x,y)/wg", "lds", "vgpr", "sgpr", "calls",
                                                            "avg_us", "min_us", "max_us", "%time"))
    for n, gx, gy, wx, lds, vg, sg, cnt, avg, mn, mx, tot in rows[:25]:
        print("%-58s %-14s %5d %6d %5d %6d %10.2f %10.2f %10.2f %6.2f" % (
            short(n)[:58], "%dx%d/%d" % (gx // max(wx, 1), gy, wx), lds, vg, sg, cnt, avg / 1e3, mn / 1e3, mx / 1e3,
            100.0 * tot / total))


def pmc(dbs):
    for db in dbs:
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, grid_size_x, grid_size_y, workgroup_size_x, counter_name, count(*), "
                         "avg(value), min(value), max(value) from counters_collection "
                         "group by kernel_name, grid_size_x, grid_size_y, counter_name order by avg(value) desc").fetchall()
        print("%-58s %-14s %-14s %6s %14s %14s %14s" % ("kernel", "grid(x,y)/wg", "counter", "calls", "avg", "min", "max"))
        for n, gx, gy, wx, cn, cnt, avg, mn, mx in rows[:60]:
            print("%-58s %-14s %-14s %6d %14.1f %14.1f %14.1f" % (short(n)[:58], "%dx%d/%d" % (gx // max(wx, 1), gy, wx),
                                                                cn, cnt, avg, mn, mx))
        # matrix-pipe utilisation where the counters are there: MFMA-busy cycles summed over the 1024 SIMDs /
        # (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 * 1024)  -- rocprofv3's MfmaUtil, per kernel
        vals = {}
        for n, gx, gy, wx, cn, cnt, avg, mn, mx in rows:
            vals.setdefault((short(n), gx, gy, wx), {})[cn] = avg
        for (n, gx, gy, wx), v in vals.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
                print("MfmaUtil %-50s %-14s %5.1f %%   (%.0f busy cycles per SIMD of %.0f)" % (
                    n[:50], "%dx%d/%d" % (gx // max(wx, 1), gy, wx),
                    100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
                    v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0, v["GRBM_GUI_ACTIVE"] / 8.0))


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        pmc(sys.argv[2:])
