"""In-tree build of libmsda_hip.so (hipcc, gfx950).  Cross-compiles without a GPU."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "lib", "libmsda_hip.so")


def build(force=False, verbose=False):
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libmsda_hip.so failed:\n" + (res.stdout or "") + (res.stderr or ""))
    if not os.path.exists(LIB):
        raise RuntimeError("build finished but %s is missing" % LIB)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
