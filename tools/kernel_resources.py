#!/usr/bin/env python
"""Registers, scratch and LDS of every kernel of a .hip translation unit, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; device-only compile, nothing is written):

    python tools/kernel_resources.py uninext_amd/csrc/msda_fwd.hip [more.hip ...]

A kernel that lives at its occupancy limit (msda_fwd_lg3: 64 VGPRs for 8 waves per SIMD) pays for three more live values
with scratch traffic on its critical path -- tests/test_kernel_resources_cpu.py pins the numbers that matter."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-strict-aliasing", "-Wno-unused-parameter"]


def resources(path):
    """{demangled kernel name: {"vgprs": .., "agprs": .., "sgprs": .., "scratch": .., "lds": .., "occupancy": ..}}"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    res = subprocess.run([hipcc] + FLAGS + ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.abspath(path), "-o", os.devnull],
                         capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(path)))
    if res.returncode != 0:
        raise RuntimeError(res.stderr[-2000:])
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch", "LDS Size [bytes/block]": "lds",
            "Occupancy [waves/SIMD]": "occupancy", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}
    for line in res.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    names = list(out)
    if names:
        try:
            dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        except OSError:
            dem = []
        if len(dem) == len(names):
            out = {d.split("(")[0].replace("void ", ""): out[n] for d, n in zip(dem, names)}
    return out


def main():
    for path in sys.argv[1:]:
        print(path)
        for name, r in resources(path).items():
            print("  %-58s vgprs %3d  scratch %3d B/lane  spills v%d s%d  sgprs %3d  occupancy %d" % (
                name[:58], r.get("vgprs", -1), r.get("scratch", -1), r.get("vgpr_spill", 0), r.get("sgpr_spill", 0), r.get("sgprs", -1),
                r.get("occupancy", -1)))


if __name__ == "__main__":
    main()
