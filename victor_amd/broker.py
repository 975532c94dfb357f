"""One GPU, many one-point callers: an owner process that turns the chains' single-point calls into batched launches.

cobaya asks the likelihood for ONE point per call (reference: ``victor/likelihoods/CCFLikelihood.py:32-39``) and gets more
throughput from several chains, one process each, under ``mpirun`` (``README.md:30``).  P such processes with a GPU context
each put P small launches in a row on the device (one point is ~12 us of kernel plus ~8 us of launch and hand-back) - and a
GPU box lets only a handful of processes open the device at all.  Here ONE process, the *broker*, owns the context:

* the chains attach to an array of mailboxes in shared memory (a file in ``/dev/shm``; layout: ``vk_mailbox`` in
  ``include/victor_hip.h``), write their parameter row and bump a sequence word;
* the broker sits in ``vk_serve_mailboxes`` (native loop, ``victor_amd/csrc/victor_hip.hip``): whatever is pending becomes ONE
  ``vk_eval_batch`` - chains that run in lock-step share a launch - and every mailbox gets its ``(lnL, chi2)`` back;
  it holds a few contexts (streams) and keeps one launch in flight on each, so a round's requests start at once while earlier
  rounds are still on the GPU - the launches overlap there the way the launches of separate processes do;
* the chains never load the HIP library or touch the GPU: 16 chains are one GPU process.

Nothing changes for the user's YAML or the plug-in: the route is chosen by the environment,

``VICTOR_HIP_BROKER=auto``
    the first chain to arrive starts the broker for its (model, data) configuration as a child process (before anything in
    that chain touches the GPU) and every chain of the job attaches to it; the broker exits a few seconds after the last
    chain has gone.  ``mpirun -n 16 -x VICTOR_HIP_BROKER=auto cobaya-run config/boss_cobaya_config.yaml``.
``VICTOR_HIP_BROKER=<name>``
    attach to a broker somebody started: ``python -m victor_amd.broker --config config/boss_cobaya_config.yaml --name <name>``.

(``VICTOR_HIP_BROKER_GPUS=G``: with several GPUs per node, chain ``local_rank`` uses broker / device ``local_rank % G``;
``VICTOR_HIP_DEVICE`` pins one.)  Only the plain ``CCFFit.log_likelihood(params)`` call travels through the mailbox - what
cobaya's ``calculate`` makes; anything else (keyword overrides, theory vectors, batches) is evaluated on a context of the
calling process as before.  Results are bit-identical to the single-process values: the same kernels run on the same rows
(the per-point arithmetic of every kernel is independent of the batch around it).

Memory ordering: the mailbox protocol needs "row before sequence word" and "results before sequence word".  The native side
uses release / acquire accesses; the Python client relies on x86-64's total store order for its plain stores (the GPU boxes
and the development container are x86-64; on another architecture use a native client).
"""

import ctypes as C
import fcntl
import hashlib
import json
import mmap
import os
import struct
import sys
import time

from . import _native as N
from .utils import InputError

MAGIC = b"VKBROKR1"
VERSION = 1
HEADER_BYTES = 4096
BOX_BYTES = C.sizeof(N.vk_mailbox)
STARTING, READY, FAILED, STOPPED = 0, 1, 2, 3
SHM_DIR = "/dev/shm"


class _Header(C.Structure):
    _fields_ = [
        ("magic", C.c_char * 8), ("version", C.c_uint32), ("n_slots", C.c_uint32),
        ("state", C.c_uint32), ("stop", C.c_uint32), ("server_pid", C.c_int64), ("device", C.c_int32), ("pad0", C.c_uint32),
        ("created", C.c_double), ("heartbeat", C.c_double),
        ("digest", C.c_char * 64),
        ("stats", N.vk_serve_stats),
        ("gather_window_us", C.c_double),
        ("depth", C.c_uint32), ("max_batch", C.c_uint32), ("threads", C.c_uint32), ("pad1", C.c_uint32),
        ("error", C.c_char * 512),
    ]


assert C.sizeof(_Header) <= HEADER_BYTES


def config_digest(model, data):
    """sha256 over the canonical JSON of the (model, data) option blocks: what a broker serves and what a chain asks for."""
    blob = json.dumps({"model": model, "data": data}, sort_keys=True, default=str).encode()
    return hashlib.sha256(blob).hexdigest()


def shm_path(name):
    if not name or "/" in name:
        raise InputError(f"bad broker name {name!r}")
    return os.path.join(SHM_DIR, name)


def auto_name(digest, device=0):
    return f"victor_broker_{os.getuid()}_{digest[:16]}_gpu{int(device)}"


def broker_device(environ=None):
    """GPU (and broker) of this chain: VICTOR_HIP_DEVICE, else the launcher's local rank modulo VICTOR_HIP_BROKER_GPUS (default 1).
    Deliberately without asking the HIP runtime how many devices there are - a chain must not initialise the GPU."""
    env = os.environ if environ is None else environ
    if env.get("VICTOR_HIP_DEVICE", "") != "":
        return int(env["VICTOR_HIP_DEVICE"])
    gpus = max(int(env.get("VICTOR_HIP_BROKER_GPUS", "1") or 1), 1)
    from .rendezvous import launcher_ranks
    found = launcher_ranks(env)
    return (found[2] % gpus) if found else 0


def _pid_alive(pid):
    if pid <= 0:
        return False
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    # a zombie (a dead child nobody has waited for yet) still answers kill(0)
    try:
        with open(f"/proc/{pid}/stat") as fh:
            return fh.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return True


class _Segment:
    """The mapped file: header + mailboxes."""

    def __init__(self, path, create=False, n_slots=0):
        self.path = path
        if create:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            try:
                os.ftruncate(fd, HEADER_BYTES + n_slots * BOX_BYTES)
            except OSError:
                os.close(fd)
                os.unlink(path)
                raise
        else:
            fd = os.open(path, os.O_RDWR)
        try:
            size = os.fstat(fd).st_size
            if size < HEADER_BYTES + BOX_BYTES:
                raise InputError(f"{path} is not a broker segment")
            self.mm = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        self.header = _Header.from_buffer(self.mm)
        if create:
            self.header.magic = MAGIC
            self.header.version = VERSION
            self.header.n_slots = n_slots
            self.header.state = STARTING
            self.header.created = time.time()
        self.n_slots = (size - HEADER_BYTES) // BOX_BYTES
        self.boxes = (N.vk_mailbox * self.n_slots).from_buffer(self.mm, HEADER_BYTES)

    def close(self):
        # ctypes views keep the buffer exported: drop them before the map
        self.header = None
        self.boxes = None
        try:
            self.mm.close()
        except BufferError:
            pass


# ======================================================================================================================
# the owner
# ======================================================================================================================
class Broker:
    """Owns the GPU context of one (model, data) configuration and serves the mailboxes of segment ``name``.

    ``attach_existing``: the segment was created (state STARTING) by the chain that elected itself to start this process.
    """

    def __init__(self, model, data, name, n_slots=64, device=0, gather_window_us=3.0, attach_existing=False, depth=4, max_batch=8,
                 threads=1, digest=None):
        """``depth``: contexts (streams) per serving thread = launches it may have in flight at once (1..8); ``max_batch``:
        requests per launch (0: the library's limit, 32); ``threads``: serving threads, each with its own contexts and its own
        contiguous share of the mailboxes.  Defaults from tools/gpu_broker_sweep.py (profiles/r04/broker_sweep.txt,
        broker_sweep_threads.txt; max_batch 4 -> 8 with the polling hand-off, broker_sweep_poll.txt: 16 chains 505 -> 520-600 k
        evaluations/s, 4 and 8 chains unchanged): more threads or more contexts buy nothing - 8 chains saturate near 2.6e5 evaluations/s whether
        they travel one per launch on 4 threads x 2 contexts or four per launch on one thread, because what is short is the GPU's
        latency for eight single-point work splits in flight at once, not the host's time to enqueue them."""
        self.name = name
        depth = max(1, min(int(depth), 8))
        threads = max(1, min(int(threads), 8))
        path = shm_path(name)
        self.seg = _Segment(path, create=not attach_existing, n_slots=n_slots)
        h = self.seg.header
        if attach_existing and (bytes(h.magic) != MAGIC or h.state != STARTING):
            raise InputError(f"{path} is not a broker segment waiting for its server")
        h.server_pid = os.getpid()
        h.device = int(device)
        h.gather_window_us = float(gather_window_us)
        # (an owner started by a chain is told that chain's digest: its own copy of the blocks came through a JSON file)
        h.digest = (digest or config_digest(model, data)).encode()
        self.fit = None
        try:
            from .ccf_fit import CCFFit
            self.fit = CCFFit(model, data, device=device, broker=False)
            plan = self.fit._single_point_plan()
            if plan is None:
                raise InputError("beta_interpolation 'likelihood' evaluates two rows per point: not served by the broker")
            self.engine, self.opts = plan[0], plan[4]
            # one context (stream, device tables, pinned buffers) per launch in flight
            from .engine import Engine
            key = self.fit._engine_key(self.fit._merged({}))
            threads = min(threads, self.seg.n_slots)
            self.engines = [self.engine] + [Engine(self.fit, self.fit, device=device, matter_model=key,
                                                   simpson_even=self.engine.simpson_even) for _ in range(depth * threads - 1)]
            # first evaluation here, not under the first client's clock (runtime, code object, LDS image)
            for eng in self.engines:
                eng.eval_point(plan[1], [0.5, 380.0, 1.0, 1.0, 1.0, 0.4, 1.0, 1.0, 1.0, float(self.fit.model["bias"]), 0.0, 0.0])
            h.depth = depth
            h.threads = threads
            h.max_batch = max(0, int(max_batch))
        except Exception as exc:
            h.error = str(exc).encode()[:500]
            h.state = FAILED
            raise
        h.heartbeat = time.time()
        h.state = READY

    def serve(self, linger=None, parent_pid=None, slice_s=0.25):
        """Serve until the header's stop word is set - or, with ``linger``, until that many seconds have passed without any
        client attached after at least one had been (``parent_pid``: also when that process has gone and nobody is attached)."""
        import threading
        seg, lib = self.seg, self.engine._lib
        h = seg.header
        stop_addr = C.addressof(h) + _Header.stop.offset
        n_thr, depth = int(h.threads) or 1, int(h.depth)
        bounds = thread_bounds(seg.n_slots, n_thr)
        stats = [N.vk_serve_stats() for _ in range(n_thr)]
        ctxs = [(C.c_void_p * depth)(*[e._ctx for e in self.engines[g * depth:(g + 1) * depth]]) for g in range(n_thr)]
        boxes_addr = C.addressof(seg.boxes)
        errors, quit_flag = [], []
        ever, empty_since = False, time.time()

        def serve(g):
            """Thread g: slices of the native loop over its mailboxes (ctypes releases the GIL for their duration)."""
            lo, hi = bounds[g]
            while not quit_flag and not errors and not h.stop:
                rc = lib.vk_serve_mailboxes(ctxs[g], depth, C.byref(self.opts), boxes_addr + lo * BOX_BYTES, hi - lo, stop_addr,
                                            float(h.gather_window_us), int(h.max_batch), float(slice_s), C.byref(stats[g]))
                if rc != 0:
                    errors.append((g, rc))

        workers = [threading.Thread(target=serve, args=(g,), daemon=True) for g in range(n_thr)]
        for t in workers:
            t.start()
        try:
            while True:
                time.sleep(slice_s)
                if errors:
                    g, rc = errors[0]
                    self.engines[g * depth]._check(rc)
                total = N.vk_serve_stats()
                for st in stats:
                    total.batches += st.batches
                    total.evals += st.evals
                    total.max_batch = max(total.max_batch, st.max_batch)
                    total.windows_timed_out += st.windows_timed_out
                    total.busy_seconds += st.busy_seconds
                h.stats = total
                now = time.time()
                h.heartbeat = now
                # (the mailbox of a chain that died without detaching is handed on by the native loop itself, at the start of
                # its next slice, when no launch carries that chain's request any more - vk_serve_mailboxes; here it just
                # does not count as a client)
                attached = sum(1 for box in seg.boxes if box.state == N_BOX_ATTACHED and _pid_alive(box.client_pid))
                if attached:
                    ever, empty_since = True, now
                if h.stop:
                    break
                if linger is not None and ever and now - empty_since > linger:
                    break
                if parent_pid and not attached and not _pid_alive(parent_pid) and now - empty_since > (linger or 0.0):
                    break
        finally:
            quit_flag.append(1)
            h.stop = 1
            for t in workers:
                t.join(timeout=5)
            h.state = STOPPED
            self.close()

    def close(self):
        path = self.seg.path
        self.seg.close()
        for p in (path, path + ".lock"):
            try:
                os.unlink(p)
            except OSError:
                pass


N_BOX_FREE, N_BOX_ATTACHED = 0, 1      # VK_BOX_* of include/victor_hip.h


def thread_bounds(n_slots, n_threads):
    """Contiguous ranges of mailboxes, one per serving thread: [(lo, hi), ...]."""
    n_threads = max(1, min(int(n_threads) or 1, n_slots))
    base, extra = divmod(n_slots, n_threads)
    out, lo = [], 0
    for g in range(n_threads):
        hi = lo + base + (1 if g < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


# ======================================================================================================================
# the chains' side
# ======================================================================================================================
class BrokerClient:
    """A chain's mailbox.  ``eval_point(row) -> (lnL, chi2)`` for one row of VK_NPAR floats."""

    def __init__(self, name, digest, timeout=300.0):
        self.name = name
        path = shm_path(name)
        deadline = time.monotonic() + timeout
        while True:
            try:
                self.seg = _Segment(path)
                break
            except (FileNotFoundError, InputError):
                if time.monotonic() > deadline:
                    raise N.NativeError(f"no broker segment at {path}")
                time.sleep(0.05)
        h = self.seg.header
        # the owner is starting (HIP runtime, tables, first evaluations): wait - but not for a process that is not there.  The
        # segment is created by the chain that elected itself BEFORE it starts the owner, which writes its pid first thing:
        # no pid within `unborn` seconds, or a pid whose process has gone (crashed or killed during HIP initialisation, before
        # it could write FAILED), ends the wait instead of holding every chain of the job for the full time-out.
        unborn = time.monotonic() + min(timeout, 60.0)
        while h.state == STARTING:
            pid = int(h.server_pid)
            if pid > 0 and not _pid_alive(pid):
                raise N.NativeError(f"broker {name}: its owner process (pid {pid}) died while starting - remove {path} or use "
                                    "VICTOR_HIP_BROKER=auto, which clears such leftovers")
            if pid <= 0 and time.monotonic() > unborn:
                raise N.NativeError(f"broker {name}: no owner process has taken the segment {path}")
            if time.monotonic() > deadline:
                raise N.NativeError(f"broker {name} did not become ready within {timeout:.0f} s")
            time.sleep(0.02)
        if h.state == FAILED:
            raise N.NativeError(f"broker {name} failed to start: {bytes(h.error).split(bytes(1))[0].decode(errors='replace')}")
        if h.state != READY or bytes(h.magic) != MAGIC or h.version != VERSION:
            raise N.NativeError(f"broker {name} is not serving (state {h.state})")
        if bytes(h.digest).decode() != digest:
            raise InputError(f"broker {name} serves another (model, data) configuration than this chain's")
        self.server_pid = int(h.server_pid)
        if not _pid_alive(self.server_pid):
            raise N.NativeError(f"broker {name}: its owner process (pid {self.server_pid}) is gone - remove {path} or use "
                                "VICTOR_HIP_BROKER=auto, which clears such leftovers")
        # claim a free mailbox; chains of one job race for them, so under a file lock
        self.slot = None
        lock_fd = os.open(path + ".lock", os.O_CREAT | os.O_RDWR, 0o600)
        try:
            fcntl.flock(lock_fd, fcntl.LOCK_EX)
            # a free mailbox in the share of the serving thread that has the fewest chains
            best, boxes = None, self.seg.boxes
            for lo, hi in thread_bounds(self.seg.n_slots, int(h.threads) or 1):
                used = sum(1 for i in range(lo, hi) if boxes[i].state == N_BOX_ATTACHED)
                free = next((i for i in range(lo, hi) if boxes[i].state == N_BOX_FREE), None)
                if free is not None and (best is None or used < best[0]):
                    best = (used, free)
            if best is not None:
                box = boxes[best[1]]
                box.req_seq = box.resp_seq = 0
                box.client_pid = os.getpid()
                box.state = N_BOX_ATTACHED
                self.slot = best[1]
        finally:
            fcntl.flock(lock_fd, fcntl.LOCK_UN)
            os.close(lock_fd)
        if self.slot is None:
            raise N.NativeError(f"broker {name}: all {self.seg.n_slots} mailboxes are taken")
        off = HEADER_BYTES + self.slot * BOX_BYTES
        self._mm = self.seg.mm
        self._off_row = off + N.vk_mailbox.row.offset
        self._words = memoryview(self._mm).cast("B")[off:off + BOX_BYTES].cast("Q")      # 32 words; [0] req_seq, [24] resp_seq
        self._off_out = off + N.vk_mailbox.lnl.offset
        self._seq = 0
        self._pid = os.getpid()
        self._pack_row = struct.Struct(f"<{N.VK_NPAR}d").pack_into
        self._unpack_out = struct.Struct("<ddi").unpack_from

    def eval_point(self, row):
        if self._pid != os.getpid():
            raise N.NativeError("a broker mailbox belongs to the process that attached it (forked child: construct a new CCFFit)")
        words = self._words
        self._pack_row(self._mm, self._off_row, *row)
        seq = self._seq = self._seq + 1
        words[0] = seq                       # after the row: x86-64 keeps the order of the two stores
        spins = 0
        while words[24] != seq:
            spins += 1
            if spins & 0x3FF == 0:           # ~every 50 us of waiting: let somebody else have the core, look after the server
                os.sched_yield()
                if spins & 0xFFFFF == 0 and not _pid_alive(self.server_pid):
                    raise N.NativeError(f"broker {self.name} (pid {self.server_pid}) has gone away")
        lnl, chi2, status = self._unpack_out(self._mm, self._off_out)
        if status != 0:
            if status == -1:
                raise InputError(f"broker {self.name}: the evaluation was refused (VK_E_ARG)")
            raise N.NativeError(f"broker {self.name}: libvictor_hip error {status}")
        return lnl, chi2

    def close(self):
        seg = getattr(self, "seg", None)
        if seg is None:
            return
        if self.slot is not None and self._pid == os.getpid() and seg.boxes is not None:
            seg.boxes[self.slot].state = N_BOX_FREE
        self._words.release()
        self._words = None
        seg.close()
        self.seg = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def spawn_broker(model, data, name, device=0, n_slots=64, linger=5.0, gather_window_us=3.0, log=None, depth=4, max_batch=8,
                 threads=1, digest=None):
    """Start ``python -m victor_amd.broker`` for a segment this process has just created (election winner) or will create.
    The child is a fresh interpreter: it is the only process that initialises the GPU."""
    import subprocess
    import tempfile
    cfg = tempfile.NamedTemporaryFile("w", suffix=".json", prefix="victor_broker_", delete=False)
    json.dump({"model": model, "data": data}, cfg, default=str)
    cfg.close()
    cmd = [sys.executable, "-m", "victor_amd.broker", "--config-json", cfg.name, "--name", name, "--device", str(device),
           "--slots", str(n_slots), "--linger", str(linger), "--window-us", str(gather_window_us), "--depth", str(depth), "--max-batch", str(max_batch), "--threads", str(threads),
           "--attach-existing", "--digest", digest or config_digest(model, data),
           "--parent-pid", str(os.getpid()), "--delete-config"]
    env = dict(os.environ)
    env.pop("VICTOR_HIP_BROKER", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    out = open(log, "ab") if log else subprocess.DEVNULL
    return subprocess.Popen(cmd, env=env, stdin=subprocess.DEVNULL, stdout=out, stderr=out if log else None,
                            start_new_session=True, cwd=os.getcwd())


def connect(model, data, spec, timeout=300.0):
    """The chain's entry point: ``spec`` is the value of VICTOR_HIP_BROKER.  Returns a :class:`BrokerClient`."""
    digest = config_digest(model, data)
    if spec != "auto":
        return BrokerClient(spec, digest, timeout=timeout)
    device = broker_device()
    name = auto_name(digest, device)
    path = shm_path(name)
    n_slots = int(os.environ.get("VICTOR_HIP_BROKER_SLOTS", "64"))
    # The election - look at what is there, clear a leftover, create the segment, start the owner - is one critical section
    # under a file lock: two chains that both find the leftover of a dead job must not remove each other's fresh segment.
    lock_fd = os.open(path + ".elect", os.O_CREAT | os.O_RDWR, 0o600)
    try:
        fcntl.flock(lock_fd, fcntl.LOCK_EX)
        for _ in range(3):
            try:
                seg = _Segment(path, create=True, n_slots=n_slots)
            except FileExistsError:
                try:
                    old = _Segment(path)
                except (FileNotFoundError, InputError):
                    time.sleep(0.05)
                    continue
                h = old.header
                dead_server = h.state in (READY, STARTING) and h.server_pid > 0 and not _pid_alive(int(h.server_pid))
                never_started = h.state == STARTING and h.server_pid == 0 and time.time() - h.created > 120.0
                stale = dead_server or never_started or h.state in (FAILED, STOPPED)
                old.close()
                if not stale:
                    break                                   # somebody else's owner is (or is becoming) ready: attach below
                for p in (path, path + ".lock"):
                    try:
                        os.unlink(p)
                    except OSError:
                        pass
                continue
            seg.close()
            spawn_broker(model, data, name, device=device, n_slots=n_slots, log=os.environ.get("VICTOR_HIP_BROKER_LOG"),
                         depth=int(os.environ.get("VICTOR_HIP_BROKER_DEPTH", "4")),
                         max_batch=int(os.environ.get("VICTOR_HIP_BROKER_MAX_BATCH", "8")),
                         threads=int(os.environ.get("VICTOR_HIP_BROKER_THREADS", "1")), digest=digest)
            break
        else:
            raise N.NativeError(f"could not start or reach broker {name}")
    finally:
        fcntl.flock(lock_fd, fcntl.LOCK_UN)
        os.close(lock_fd)
    return BrokerClient(name, digest, timeout=timeout)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="GPU owner process for many single-point likelihood callers (see module docstring)")
    ap.add_argument("--config", help="YAML file with model / data blocks (config/boss_config.yaml) or a cobaya file "
                                     "(likelihood: CCFLikelihood: {model, data})")
    ap.add_argument("--config-json", help="JSON file {model, data} (written by spawn_broker)")
    ap.add_argument("--name", help="segment name in /dev/shm (default: derived from the configuration)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--slots", type=int, default=64)
    ap.add_argument("--linger", type=float, default=None, help="exit this many seconds after the last client has gone")
    ap.add_argument("--window-us", type=float, default=3.0, help="how long a round waits for the other chains' requests")
    ap.add_argument("--depth", type=int, default=4, help="contexts (streams) of the owner = launches in flight at once")
    ap.add_argument("--threads", type=int, default=1, help="serving threads, each with `depth` contexts and its share of the mailboxes")
    ap.add_argument("--max-batch", type=int, default=8, help="requests per launch (0: the library's limit of 32); 8 BOSS "
                                                             "requests are the most one launch hands over by polling "
                                                             "(tools/gpu_broker_sweep.py, profiles/r04/broker_sweep_poll.txt)")
    ap.add_argument("--attach-existing", action="store_true")
    ap.add_argument("--digest", default=None, help="configuration digest to publish (set by the chain that starts the owner)")
    ap.add_argument("--parent-pid", type=int, default=0)
    ap.add_argument("--delete-config", action="store_true")
    args = ap.parse_args(argv)
    if args.config_json:
        with open(args.config_json) as fh:
            info = json.load(fh)
        if args.delete_config:
            os.unlink(args.config_json)
    elif args.config:
        import yaml
        with open(args.config) as fh:
            info = yaml.full_load(fh)
        if "likelihood" in info:
            info = next(iter(info["likelihood"].values()))
    else:
        ap.error("--config or --config-json is required")
    model, data = info["model"], info["data"]
    name = args.name or auto_name(config_digest(model, data), args.device)
    try:
        broker = Broker(model, data, name, n_slots=args.slots, device=args.device, gather_window_us=args.window_us,
                        attach_existing=args.attach_existing, depth=args.depth, max_batch=args.max_batch, threads=args.threads,
                        digest=args.digest)
    except Exception as exc:
        if args.attach_existing:            # tell the chains that are waiting for READY
            try:
                seg = _Segment(shm_path(name))
                seg.header.error = str(exc).encode()[:500]
                seg.header.state = FAILED
                seg.close()
            except Exception:
                pass
        raise
    import signal

    def leave(signum, frame):            # SIGTERM / SIGINT: finish what is in flight, remove the segment, go
        broker.seg.header.stop = 1

    signal.signal(signal.SIGTERM, leave)
    signal.signal(signal.SIGINT, leave)
    print(f"victor broker '{name}' ready: pid {os.getpid()}, device {args.device}, {broker.seg.n_slots} mailboxes, "
          f"{int(broker.seg.header.threads)} serving thread(s) x {int(broker.seg.header.depth)} launches in flight", file=sys.stderr, flush=True)
    broker.serve(linger=args.linger, parent_pid=args.parent_pid or None)


if __name__ == "__main__":
    main()
