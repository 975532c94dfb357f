"""Single-rank RCCL communicator through the C ABI with RCCL's own logging on: which HIP call fails, with which library.
Usage: gpu_rccl_diag.py            (VICTOR_HIP_DEV=1 VICTOR_HIP_RCCL_LIB=... selects another librccl; NCCL_DEBUG etc. are honoured)"""
import os, sys
os.environ.setdefault("NCCL_DEBUG", "INFO")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from victor_amd import _native
from tests import cases

print("env HSA_ENABLE_IPC_MODE_LEGACY =", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), flush=True)
fit = victor_amd.CCFFit(*cases.synth_options(2))
eng = fit._get_engine()
print(_native.comm_info(), flush=True)
n = 1024
d_send, d_recv = eng.alloc(n), eng.alloc(n)
eng.upload(d_send, np.arange(n, dtype=float))
uid = eng.comm_unique_id()
eng.comm_init(uid, 0, 1)
print("communicator built", flush=True)
try:
    eng.comm_allgather_async(d_send, d_recv, n)
    eng.sync()
    out = eng.download(d_recv, n)
    print("allgather ok:", bool(np.array_equal(out, np.arange(n))), flush=True)
except Exception as exc:
    print("allgather FAILED:", exc, flush=True)
eng.comm_destroy()
