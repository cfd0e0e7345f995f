#!/usr/bin/env python
"""Where a kernel spills: compiles one .hip with -g1 -save-temps and lists scratch / v_readlane / v_writelane instructions per source line.

    python tools/spill_lines.py uninext_amd/csrc/msda_bwd_win2.hip msda_bwd_win2 [extra hipcc flags]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile


def main():
    src, kern = os.path.abspath(sys.argv[1]), sys.argv[2]
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-strict-aliasing",
                        "-Wno-unused-parameter", "-g1", "-save-temps", "-I", os.path.dirname(src), "-c", src, "-o", "x.o"] + sys.argv[3:],
                       cwd=td, check=True, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
        lines = open(os.path.join(td, asm)).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kern, l)]
    for st in starts:
        end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
        cur, cnt, total = None, collections.Counter(), 0
        for l in lines[st:end]:
            m = re.search(r"\.loc\s+\d+\s+(\d+)", l)
            if m:
                cur = int(m.group(1))
            t = l.split()
            if t and not t[0].startswith((".", ";")) and not t[0].endswith(":"):
                total += 1
            if "scratch_" in l or "v_readlane" in l or "v_writelane" in l:
                cnt[(cur, t[0])] += 1
        print(lines[st].rstrip(":"), total, "instructions")
        for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0] or 0, kv[0][1])):
            print("  line %-5s %-24s %d" % (k[0], k[1], v))


if __name__ == "__main__":
    main()
