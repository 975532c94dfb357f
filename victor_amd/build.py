"""Compile ``csrc/victor_hip.hip`` into ``csrc/libvictor_hip.so`` for gfx950 with hipcc (in-tree)."""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "victor_hip.hip")
OUT = os.path.join(HERE, "csrc", "libvictor_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_native(force=False, verbose=False):
    import glob
    deps = [SRC, os.path.join(INCLUDE, "victor_hip.h")] + glob.glob(os.path.join(HERE, "csrc", "*.h"))
    if not force and os.path.isfile(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", INCLUDE, "-o", OUT, SRC, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
