"""Compile ``csrc/*.hip`` into ``csrc/libvictor_hip.so`` for gfx950 with hipcc (in-tree).

Two translation units, compiled side by side and linked into one library:

``victor_hip.hip``           the C ABI, the host side and every other kernel;
``vk_cells_streaming.hip``   the cells kernel's instantiations for the streaming model (the kernels of the headline
                         metric and of the BOSS configuration), compiled with LLVM's ``iterative-ilp`` machine scheduler: it
                         interleaves the independent chains of the node loop and fills the hazard slots the default
                         (occupancy-driven) scheduler leaves as ``s_nop`` - about 1 % per launch, same bits (DESIGN.md
                         section 5).  The flag is per translation unit, hence the unit.
"""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "victor_hip.hip")
UNITS = (("victor_hip.hip", ()), ("vk_cells_streaming.hip", ("-mllvm", "-amdgpu-sched-strategy=iterative-ilp")))
OUT = os.path.join(CSRC, "libvictor_hip.so")
# development build: the product plus the lanes-over-the-batch yardstick kernel (vk_kernel_lanes.h, -DVK_DEV_LANES) that tools/ and
# the mapping tests compare against; never loaded by the package itself (tests/devlib.py, VICTOR_HIP_LIB)
DEV_OUT = os.path.join(CSRC, "libvictor_hip_dev.so")
DEV_DEFINES = ("-DVK_DEV_LANES",)
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
COMMON = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC")


def sources_digest():
    """sha256 over the kernel sources (csrc/*.h, csrc/*.hip, include/victor_hip.h; names and contents, sorted): what a profile was
    taken at.  tools/update_traffic.py stores it with the counters, bench.py recomputes it (`traffic_profiled.sources_unchanged`)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(INCLUDE, "victor_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_native(force=False, verbose=False, defines=(), out=None, dev=False):
    import glob
    if dev:
        defines, out = tuple(defines) + DEV_DEFINES, out or DEV_OUT
    out = out or OUT
    deps = [os.path.join(CSRC, u) for u, _ in UNITS] + [os.path.join(INCLUDE, "victor_hip.h")] + glob.glob(os.path.join(CSRC, "*.h"))
    if not force and os.path.isfile(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    hipcc = hipcc_path()
    tag = os.path.splitext(os.path.basename(out))[0]
    jobs = []
    for unit, extra in UNITS:
        obj = os.path.join(CSRC, f"{tag}.{os.path.splitext(unit)[0]}.o")
        cmd = [hipcc, *COMMON, *defines, *extra, "-I", INCLUDE, "-c", os.path.join(CSRC, unit), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((cmd, obj, subprocess.Popen(cmd)))
    failed = [(cmd, proc.returncode) for cmd, _, proc in jobs if proc.wait() != 0]      # every compile is waited for
    if failed:
        for _, obj, _ in jobs:             # no object of a failed build stays behind for a later link to pick up
            if os.path.exists(obj):
                os.remove(obj)
        raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *[obj for _, obj, _ in jobs], "-ldl"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    for _, obj, _ in jobs:
        os.remove(obj)
    return out


if __name__ == "__main__":
    import sys
    print(build_native(force=True, verbose=True, dev="--dev" in sys.argv))
