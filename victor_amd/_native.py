"""ctypes binding of ``libvictor_hip.so`` (declared in ``include/victor_hip.h``).

There is deliberately no fallback: if the shared library has not been built, or no
GPU is usable, every evaluation entry point raises.  Build with::

    python -c "import __graft_entry__ as g; g.build()"      # or: make -C victor_amd/csrc
"""

import ctypes as C
import os

import numpy as np

VK_ABI_VERSION = 18
VK_NPAR = 12
(P_FSIGMA8, P_SIGMAV, P_APERP, P_APAR, P_EPSILON, P_BETA, P_ASTAR, P_M, P_Q, P_BIAS, P_AV, P_SPARE) = range(12)
MATTER = {"template": 0, "linear_bias": 1, "velocity_template": 2}
RSD = {"streaming": 0, "dispersion": 1, "kaiser": 2, "euclid_special": 3}
LIKE = {"gaussian": 0, "sellentin": 1, "hartlap": 2, "percival": 3}
VK_COMM_ID_BYTES = 128
VK_WALK_EPSILON = -1

_dp = C.POINTER(C.c_double)


class vk_pp(C.Structure):
    _fields_ = [("n_int", C.c_int32), ("lead", C.c_int32), ("inv_h", C.c_double),
                ("knots", _dp), ("coef", _dp)]


class vk_tables(C.Structure):
    _fields_ = [
        ("n_s", C.c_int32), ("n_mu", C.c_int32), ("n_x", C.c_int32), ("n_ell", C.c_int32),
        ("s", _dp), ("mu", _dp), ("w_ell", _dp), ("x", _dp), ("w_x", _dp),
        ("n_ell_r", C.c_int32), ("n_beta_r", C.c_int32), ("beta_r", _dp), ("xi", vk_pp),
        ("matter_model", C.c_int32), ("vr_beta_dep", C.c_int32), ("vr", vk_pp), ("vr_emp", C.POINTER(C.c_double)),
        ("vt_amp", C.c_double),
        ("sv", vk_pp), ("sv_n_mu", C.c_int32), ("sv_mu_inv_h", C.c_double), ("sv_mu", _dp), ("sv2d", _dp),
        ("uni_n", C.c_int32), ("uni_u0", C.c_double), ("uni_inv_h", C.c_double), ("uni_sv_v", _dp), ("uni_xi", _dp), ("uni_xic", _dp), ("uni_vb", _dp), ("uni_v2", _dp), ("uni_da", _dp), ("uni_ge", _dp), ("uni_dab", _dp), ("uni_empb", _dp),
        ("uni_lut_n", C.c_int32), ("uni_lut_inv_g", C.c_double), ("uni_lut", C.POINTER(C.c_uint16)), ("uni_knots", _dp),
        ("iaH", C.c_double), ("template_sigma8", C.c_double),
        ("n_beta_d", C.c_int32), ("beta_d", _dp), ("data", _dp),
        ("n_beta_c", C.c_int32), ("beta_c", _dp), ("prec", _dp), ("logdet", _dp), ("eig", _dp),
    ]


class vk_eval_opts(C.Structure):
    _fields_ = [
        ("rsd_model", C.c_int32), ("assume_isotropic", C.c_int32), ("rescale_from_ap", C.c_int32),
        ("like_form", C.c_int32), ("nmocks", C.c_double), ("nparams", C.c_double),
        ("kaiser_approx", C.c_int32), ("kaiser_coord_shift", C.c_int32), ("niter", C.c_int32),
        ("from_data", C.c_int32), ("empirical_corr", C.c_int32), ("reserved", C.c_int32),
    ]


class vk_mailbox(C.Structure):
    """One caller's slot in the shared-memory array ``vk_serve_mailboxes`` serves (include/victor_hip.h; 256 bytes)."""
    _fields_ = [
        ("req_seq", C.c_uint64), ("state", C.c_uint32), ("reserved0", C.c_uint32), ("client_pid", C.c_int64),
        ("reserved1", C.c_uint64 * 5),
        ("row", C.c_double * VK_NPAR),
        ("reserved2", C.c_uint64 * 4),
        ("resp_seq", C.c_uint64), ("lnl", C.c_double), ("chi2", C.c_double), ("status", C.c_int32), ("reserved3", C.c_uint32),
        ("reserved4", C.c_uint64 * 4),
    ]


class vk_serve_stats(C.Structure):
    _fields_ = [("batches", C.c_uint64), ("evals", C.c_uint64), ("max_batch", C.c_uint64), ("windows_timed_out", C.c_uint64),
                ("busy_seconds", C.c_double)]


assert C.sizeof(vk_mailbox) == 256 and vk_mailbox.row.offset == 64 and vk_mailbox.resp_seq.offset == 192


class NativeError(RuntimeError):
    """The HIP library is missing or a device call failed."""


class CommInitTimeout(NativeError):
    """``ncclCommInitRank`` did not return: a helper thread is still inside RCCL on this context, so the context (and with it
    the process's GPU state) must not be used again.  Fatal for the process - report it and exit non-zero so that the job is
    restarted as fresh processes; do not fall back to another gather on the same context."""


def library_path():
    env = os.environ.get("VICTOR_HIP_LIB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libvictor_hip.so")


# every symbol include/victor_hip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_optp = C.POINTER(vk_eval_opts)
SYMBOLS = {
    "vk_abi_version": (C.c_int, []),
    "vk_device_count": (C.c_int, []),
    "vk_knobs_refresh": (None, []),
    "vk_poll_rule": (C.c_int32, [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "vk_poll_grant": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "vk_poll_device_reserved": (C.c_int32, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vk_poll_budget": (C.c_int32, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vk_create": (_vp, [C.POINTER(vk_tables), C.c_int, C.c_char_p, C.c_size_t]),
    "vk_destroy": (None, [_vp]),
    "vk_last_error": (C.c_char_p, [_vp]),
    "vk_last_kernel": (C.c_char_p, [_vp]),
    "vk_last_fused": (C.c_int, [_vp]),
    "vk_last_polled": (C.c_int, [_vp]),
    "vk_default_opts": (None, [_optp]),
    "vk_eval_batch": (C.c_int, [_vp, _optp, _dp, C.c_int64, _dp, _dp, _dp]),
    "vk_eval_batch_begin": (C.c_int, [_vp, _optp, _dp, C.c_int64]),
    "vk_eval_batch_finish": (C.c_int, [_vp, _dp, _dp]),
    "vk_walk_create": (_vp, [C.POINTER(C.c_void_p), C.c_int32, _optp, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _dp, _dp, _dp,
                             C.c_double, C.c_int32, C.c_char_p, C.c_size_t]),
    "vk_walk_run": (C.c_int, [_vp, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vk_walk_last_error": (C.c_char_p, [_vp]),
    "vk_walk_destroy": (None, [_vp]),
    "vk_epsilon_to_ap": (None, [_dp, C.c_int64, C.c_double, _dp, _dp]),
    "vk_theory_batch": (C.c_int, [_vp, _optp, _dp, C.c_int64, _dp, C.c_int32, _dp, C.c_int32, _dp, C.c_int32, _dp]),
    "vk_xi_smu_batch": (C.c_int, [_vp, _optp, _dp, C.c_int64, _dp, C.c_int32, _dp, C.c_int32, _dp]),
    "vk_device_alloc": (_vp, [_vp, C.c_size_t]),
    "vk_device_free": (None, [_vp, _vp]),
    "vk_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "vk_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "vk_eval_batch_device_async": (C.c_int, [_vp, _optp, _vp, C.c_int64, _vp, _vp, _vp]),
    "vk_sync": (C.c_int, [_vp]),
    "vk_joint_workspace_doubles": (C.c_size_t, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64]),
    "vk_joint_eval_device_async": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, _optp, _vp, C.c_int64, _vp, _vp, _vp]),
    "vk_serve_mailboxes": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, _optp, _vp, C.c_int32, _vp, C.c_double, C.c_int32,
                           C.c_double, C.POINTER(vk_serve_stats)]),
    "vk_timing_enable": (C.c_int, [_vp, C.c_int]),
    "vk_timing_read": (C.c_int, [_vp, _dp, _dp, C.POINTER(C.c_int64), C.c_int]),
    "vk_comm_unique_id": (C.c_int, [C.c_char_p]),
    "vk_comm_init": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    "vk_comm_allgather_async": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "vk_comm_allgather_host_begin": (C.c_int, [_vp, _dp, C.c_int64]),
    "vk_comm_allgather_host_finish": (C.c_int, [_vp, _dp]),
    "vk_comm_destroy": (C.c_int, [_vp]),
    "vk_device_bus_id": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "vk_comm_init_all": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32]),
    "vk_comm_allgather_group_async": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int64]),
    "vk_comm_info": (C.c_int, [C.c_char_p, C.c_size_t]),
    "vk_comm_rank_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    # development entry points of the polling ledger (answer only with VICTOR_HIP_DEV=1; tests/test_ledger.py)
    "vk_ledger_layout": (None, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vk_ledger_self": (C.c_uint64, [C.c_int32, C.c_int64]),
    "vk_ledger_open_at": (_vp, [C.c_char_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_int32)]),
    "vk_ledger_slot": (C.c_int32, [_vp]),
    "vk_ledger_others": (C.c_int32, [_vp]),
    "vk_ledger_generation": (C.c_uint32, [_vp]),
    "vk_ledger_grant": (C.c_int32, [_vp, C.POINTER(C.c_int32), C.c_int32, C.c_int32]),
    "vk_ledger_release": (None, [_vp, C.POINTER(C.c_int32), C.c_int32]),
    "vk_ledger_close": (None, [_vp, C.c_int32]),
}

_lib = None
_loaded = {}          # path -> CDLL: the product library and, in development runs, other builds of it (load_path)


def load_path(path):
    """Load one build of the library (once per path) and declare every prototype."""
    path = os.path.abspath(path)
    if path in _loaded:
        return _loaded[path]
    if not os.path.isfile(path):
        raise NativeError(
            f"{path} not found: the HIP extension has not been built "
            "(run `python -c \"import __graft_entry__ as g; g.build()\"`). There is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as exc:
        raise NativeError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.vk_abi_version() != VK_ABI_VERSION:
        raise NativeError(f"{path}: ABI version mismatch; rebuild it")
    _loaded[path] = lib
    return lib


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is None:
        _lib = load_path(library_path())
    return _lib


def set_knob(name, value):
    """Set (``value`` a string) or clear (``None``) a ``VICTOR_HIP_*`` tuning / A-B knob in this process and make every
    context re-read the knobs at its next call (they are cached per context, not read per launch).  Development only: the
    library ignores every knob unless ``VICTOR_HIP_DEV=1`` is set, which this function does (tests/ and tools/ go through
    it); a production process that merely inherits a ``VICTOR_HIP_*`` variable runs the default kernels."""
    os.environ["VICTOR_HIP_DEV"] = "1"
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    for lib in _loaded.values():
        lib.vk_knobs_refresh()


def comm_info():
    """Which HIP runtime / RCCL the process is using (dict; see vk_comm_info in include/victor_hip.h)."""
    import json
    buf = C.create_string_buffer(2048)
    rc = load().vk_comm_info(buf, len(buf))
    info = json.loads(buf.value.decode() or "{}")
    info["rccl_loaded"] = rc == 0
    return info


def epsilon_to_ap(eps, alpha=1.0, aperp=None, apar=None):
    """(aperp, apar) from an array of epsilons through the library's one routine (vk_epsilon_to_ap: libm's pow, as the
    reference's Python floats take it) - also what the walkers' native step loop uses, so every route forms the same rows."""
    eps = f64(eps)
    aperp = np.empty_like(eps) if aperp is None else aperp
    apar = np.empty_like(eps) if apar is None else apar
    load().vk_epsilon_to_ap(as_dp(eps), eps.size, float(alpha), as_dp(aperp), as_dp(apar))
    return aperp, apar


def as_dp(a):
    return a.ctypes.data_as(_dp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
