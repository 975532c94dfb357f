"""Compile ``csrc/`` into ``csrc/libvictor_hip.so`` for gfx950 (in-tree): the kernels with hipcc, the host-only parts of the
library with the host compiler, every unit side by side, linked into one library.

hipcc (device code; ``vk_instances.h`` names which kernel instantiation lives in which unit):
``victor_hip.hip``           the launch side of the C ABI (context, table upload, kernel selection), the chi-square kernels;
``vk_cells_streaming.hip``   the cells kernel's instantiations for the streaming model (the kernels of the headline
                         metric and of the BOSS configuration), compiled with LLVM's ``iterative-ilp`` machine scheduler: it
                         interleaves the independent chains of the node loop and fills the hazard slots the default
                         (occupancy-driven) scheduler leaves as ``s_nop`` - about 1 % per launch, same bits (DESIGN.md
                         section 5).  The flag is per translation unit, hence the unit;
``vk_cells_dispersion.hip``, ``vk_cells_kaiser.hip``, ``vk_fast_streaming.hip``, ``vk_fast_dispersion.hip``, ``vk_generic.hip``
                         the other instantiations of the cells, point-major and generic kernels (round 6: units of their own
                         so that a forced build is as long as its slowest unit, not as the sum).

host compiler (``g++``; HIP runtime API only, no device code - ``vk_host.h`` is what they share with ``victor_hip.hip``):
``vk_ledger.cpp``            the polling hand-off's launch rule and the device-wide ledger of reserved waiters (pure POSIX);
``vk_walk.cpp``              ``vk_walk_*``: the walkers' step loop;
``vk_serve.cpp``             ``vk_serve_mailboxes``: the GPU owner's serving loop;
``vk_rccl.cpp``              ``vk_comm_*``: RCCL through dlopen.

Objects are kept under ``csrc/obj/`` (git-ignored, not sent to the GPU box) and rebuilt when older than their source or any
header: the development twin of the library (``dev=True``) differs from the product in ``victor_hip.hip`` alone and links
the same kernel objects.
"""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "victor_hip.hip")
ILP = ("-mllvm", "-amdgpu-sched-strategy=iterative-ilp")
# (source, extra flags, takes the flavour's defines): only victor_hip.hip depends on VK_DEV_LANES
UNITS = (("victor_hip.hip", (), True), ("vk_cells_streaming.hip", ILP, False), ("vk_cells_dispersion.hip", (), False),
         ("vk_cells_kaiser.hip", (), False), ("vk_fast_streaming.hip", (), False), ("vk_fast_dispersion.hip", (), False),
         ("vk_generic.hip", (), False))
HOST_UNITS = ("vk_ledger.cpp", "vk_walk.cpp", "vk_serve.cpp", "vk_rccl.cpp")
HOST_FLAGS = ("-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-invalid-offsetof", "-D__HIP_PLATFORM_AMD__")   # (the define: HIP's headers under a compiler that is not hipcc)
OBJ = os.path.join(CSRC, "obj")
OUT = os.path.join(CSRC, "libvictor_hip.so")
# development build: the product plus the lanes-over-the-batch yardstick kernel (vk_kernel_lanes.h, -DVK_DEV_LANES) that tools/ and
# the mapping tests compare against; never loaded by the package itself (tests/devlib.py, VICTOR_HIP_LIB)
DEV_OUT = os.path.join(CSRC, "libvictor_hip_dev.so")
DEV_DEFINES = ("-DVK_DEV_LANES",)
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
COMMON = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-invalid-offsetof")   # (offsetof on vk_ctx: the cross-unit layout check of vk_create)


def sources_digest():
    """sha256 over the kernel sources (csrc/*.h, csrc/*.hip, include/victor_hip.h; names and contents, sorted): what a profile was
    taken at (csrc/*.cpp as well since round 6: the host-compiled units).  tools/update_traffic.py stores it with the counters,
    bench.py recomputes it (`traffic_profiled.sources_unchanged`)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp"))) + \
        [os.path.join(INCLUDE, "victor_hip.h")]
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


def host_compiler():
    """The host compiler of the host-only units: g++ (CXX overrides); without one, hipcc's own clang in plain C++ mode."""
    for cand in (os.environ.get("CXX"), "g++", "c++"):
        path = shutil.which(cand) if cand else None
        if path:
            return [path]
    return [hipcc_path(), "-x", "c++"]


def rocm_include():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc_path()))), "include")


def _plan(defines, dev):
    """[(command, object)] for one flavour of the library: `defines` go to every hipcc unit (a profiling build), the development
    defines to victor_hip.hip alone."""
    hipcc, cxx, rocm_inc = hipcc_path(), host_compiler(), rocm_include()
    tag = "".join(sorted(d.replace("-D", ".") for d in defines))
    plan = []
    for unit, extra, flavoured in UNITS:
        mine = tuple(defines) + (DEV_DEFINES if (dev and flavoured) else ())
        name = os.path.splitext(unit)[0] + tag + (".dev" if (dev and flavoured) else "") + ".o"
        obj = os.path.join(OBJ, name)
        plan.append(([hipcc, *COMMON, *mine, *extra, "-I", INCLUDE, "-c", os.path.join(CSRC, unit), "-o", obj], obj, os.path.join(CSRC, unit)))
    for unit in HOST_UNITS:
        obj = os.path.join(OBJ, os.path.splitext(unit)[0] + ".o")
        plan.append(([*cxx, *HOST_FLAGS, "-I", rocm_inc, "-I", INCLUDE, "-c", os.path.join(CSRC, unit), "-o", obj], obj, os.path.join(CSRC, unit)))
    return plan


def _compile(jobs, verbose):
    """Run the compile commands side by side, at most one per core; every job is waited for; a failure removes the objects of the
    failed jobs (no stale object for a later link to pick up) and raises."""
    import time
    limit = max(2, os.cpu_count() or 2)
    pending, running, failed, times = list(jobs), [], [], {}
    while pending or running:
        while pending and len(running) < limit:
            cmd, obj, _ = pending.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            tmp = obj + ".tmp"
            running.append((cmd, obj, tmp, time.perf_counter(), subprocess.Popen(cmd[:-1] + [tmp])))
        still = []
        for cmd, obj, tmp, t0, proc in running:
            rc = proc.poll()
            if rc is None:
                still.append((cmd, obj, tmp, t0, proc))
                continue
            times[os.path.basename(obj)] = time.perf_counter() - t0
            if rc == 0:
                os.replace(tmp, obj)              # an object appears only when its compile has succeeded
            else:
                failed.append((cmd, rc))
                for f in (tmp, obj):
                    if os.path.exists(f):
                        os.remove(f)
        running = still
        if running:
            time.sleep(0.05)
    if failed:
        raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
    return times


def build_native(force=False, verbose=False, defines=(), out=None, dev=False, both=False):
    """Build (if out of date, or `force`) the product library, its development twin (`dev=True`), or both in one go
    (`both=True`: the kernel objects they share are compiled once).  Returns the path of the library (of the product for `both`)."""
    import glob
    flavours = [(False, OUT), (True, DEV_OUT)] if both else [(dev, out or (DEV_OUT if dev else OUT))]
    headers = [os.path.join(INCLUDE, "victor_hip.h")] + glob.glob(os.path.join(CSRC, "*.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    os.makedirs(OBJ, exist_ok=True)
    todo, links, seen = [], [], set()
    for is_dev, target in flavours:
        plan = _plan(tuple(defines), is_dev)
        newest = max([newest_header] + [os.path.getmtime(src) for _, _, src in plan])
        # (a library newer than every source and header is current even when its objects are gone: the copy on the GPU box)
        if not force and os.path.isfile(target) and os.path.getmtime(target) >= newest:
            continue
        for job in plan:
            _, obj, src = job
            fresh = os.path.isfile(obj) and os.path.getmtime(obj) >= max(newest_header, os.path.getmtime(src))
            if (force or not fresh) and obj not in seen:
                seen.add(obj)
                todo.append(job)
        links.append((target, [obj for _, obj, _ in plan]))
    if todo:
        times = _compile(todo, verbose)
        if verbose:
            print("compile seconds per unit:", {k: round(v, 1) for k, v in sorted(times.items(), key=lambda kv: -kv[1])}, flush=True)
    for target, objs in links:
        link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target + ".tmp", *objs, "-ldl"]
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.check_call(link)
        os.replace(target + ".tmp", target)
    return flavours[0][1]


if __name__ == "__main__":
    import sys
    import time
    t0 = time.perf_counter()
    if "--phases" in sys.argv:       # profiling build with wall_clock64() phase marks in the kernels (tools/gpu_phases.py; VICTOR_HIP_LIB selects it)
        print(build_native(force=True, verbose=True, defines=("-DVK_PHASES",), out=os.path.join(CSRC, "libvictor_hip_phases.so")))
    else:
        print(build_native(force="--no-force" not in sys.argv, verbose=True, dev="--dev" in sys.argv, both="--both" in sys.argv))
    print(f"build wall time {time.perf_counter() - t0:.1f} s")
