"""Multi-GPU evaluation: shard a batch of parameter points across ranks, gather the log-likelihoods.

The likelihood path has no cross-point coupling, so scaling out is pure data
parallelism: one process per GPU, rank ``g`` of ``G`` evaluates rows ``[lo_g, hi_g)`` of the batch against
its own replica of the tables (< 1.5 MB), and the only exchange is one all-gather of ``lnL`` (and
optionally ``chi2``) per batch - 8 KiB per rank at 8 GPUs x 1024 points, latency-bound on xGMI.  The gather
runs through RCCL on the context's stream (``vk_comm_allgather_async``) when a communicator exists;
``gather="host"`` uses the ranks' socket group (:mod:`victor_amd.rendezvous`) and is what the CPU tests exercise.
A single process can drive all GPUs of a node as well (:class:`MultiGPUFit`; RCCL group calls, ``vk_comm_init_all``).

The reference has no counterpart: it relies on independent MCMC chains under ``mpirun`` (README.md:30).
"""

import os

import numpy as np


def default_device(environ=None, n_devices=None):
    """GPU for this process: ``VICTOR_HIP_DEVICE`` if set, else the launcher's local rank (torchrun, Open MPI,
    MPICH / Intel MPI, Slurm) modulo the number of visible GPUs, else 0 - so that ``mpirun -n 8 cobaya-run ...``
    (the reference's way of running several chains, README.md:30) lands one chain on each GPU."""
    env = os.environ if environ is None else environ
    if env.get("VICTOR_HIP_DEVICE", "") != "":
        return int(env["VICTOR_HIP_DEVICE"])
    for key in ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "MV2_COMM_WORLD_LOCAL_RANK",
                "SLURM_LOCALID"):
        if env.get(key, "") != "":
            rank = int(env[key])
            if n_devices is None:
                from . import _native
                n_devices = _native.load().vk_device_count()
            return rank % max(int(n_devices), 1)
    return 0


def shard_bounds(n, world, rank):
    """Contiguous, balanced split of ``n`` rows: the first ``n % world`` ranks get one extra row."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def padded_chunk(n, world):
    """Rows per rank once every rank is padded to the same count (RCCL all-gather needs equal counts)."""
    return -(-int(n) // int(world))


class Dist:
    """The ranks of a multi-process run: rank / world size from the launcher's environment (torchrun, Open MPI, MPICH,
    Slurm - :func:`victor_amd.rendezvous.launcher_ranks`) and the host-side collectives the data path needs around it
    (barriers, the RCCL unique id, a few scalars), carried by a standard-library socket group
    (:class:`victor_amd.rendezvous.SocketGroup`).  No torch, no MPI binding."""

    def __init__(self, rank=None, world=None, local_rank=None):
        from .rendezvous import launcher_ranks
        found = launcher_ranks()
        self.launched = found is not None
        r, w, l = found if found else (0, 1, 0)
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.local_rank = (l if rank is None else self.rank) if local_rank is None else local_rank
        self.group = None

    def connect(self, where=None, timeout=120.0):
        """Meet the other ranks (a collective; no-op for a single rank)."""
        from .rendezvous import SocketGroup, endpoint
        if self.group is None:
            where = where or endpoint()
            # A single-node job (its ranks meet over a Unix socket): RCCL's bootstrap sockets - the only thing it uses the
            # network for inside a node, the data travels over xGMI - go over the loopback interface unless the user has
            # chosen one: always there, never filtered.  (RCCL's own default is the first other interface, e.g. a pod's veth.)
            if where[0] == "unix" and self.world > 1:
                os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            self.group = SocketGroup(self.rank, self.world, where=where, timeout=timeout)
        return self

    def barrier(self):
        if self.group is not None and self.world > 1:
            self.group.barrier()

    def broadcast_bytes(self, payload, src=0, nbytes=None):
        if self.group is None or self.world == 1:
            return payload
        return self.group.broadcast_bytes(payload, src=src)

    def max_float(self, x):
        return float(x) if self.group is None or self.world == 1 else self.group.max_float(x)

    def min_float(self, x):
        return float(x) if self.group is None or self.world == 1 else self.group.min_float(x)

    def allgather_host(self, local, chunk):
        """All-gather equal-size host arrays of ``chunk`` doubles per rank -> (world*chunk,)."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        assert local.shape == (chunk,)
        if self.group is None or self.world == 1:
            return local.copy()
        return self.group.allgather_doubles(local)

    def close(self):
        if self.group is not None:
            self.group.close()
            self.group = None


def one_device_per_rank(dist, engine):
    """True when no two ranks of the run sit on the same GPU (host name + PCI bus id, exchanged over the socket group): the
    precondition of an RCCL communicator.  A collective - every rank gets the same answer."""
    import socket
    if dist.world == 1 or dist.group is None:
        return True
    if os.environ.get("VICTOR_HIP_DEV") == "1" and os.environ.get("VICTOR_HIP_RCCL_SHARED_DEVICE_OK") == "1":
        # tests only, and like every VICTOR_HIP_* switch only under VICTOR_HIP_DEV=1: ranks sharing a GPU may build a
        # communicator of an RCCL stand-in that can live with that (tests/rccl_double, selected through VICTOR_HIP_RCCL_LIB);
        # the real RCCL refuses such a communicator itself
        return True
    mine = f"{socket.gethostname()}|{engine.bus_id()}".encode()
    ids = dist.group.allgather_bytes(mine, "devices")
    return len(set(ids)) == len(ids)


def unpad(gathered, n, world):
    """Drop the padding rows the equal-count gather added; returns the ``n`` results in batch order."""
    chunk = padded_chunk(n, world)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        parts.append(gathered[r * chunk: r * chunk + (hi - lo)])
    return np.concatenate(parts) if parts else np.empty(0)


class ShardedLikelihood:
    """Evaluate ``log_likelihood_batch`` over all ranks and return the full-batch result on every rank.

    ``evaluate(rows) -> (lnl, chi2)`` is normally ``CCFFit.log_likelihood_batch`` bound to this rank's GPU.
    """

    def __init__(self, evaluate, dist, gather="host", engine=None):
        self.evaluate = evaluate
        self.dist = dist
        self.gather = gather
        self.engine = engine
        if gather == "rccl" and engine is None:
            raise ValueError("gather='rccl' needs the rank's Engine")

    def __call__(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        n = len(rows)
        world, rank = self.dist.world, self.dist.rank
        lo, hi = shard_bounds(n, world, rank)
        chunk = padded_chunk(n, world)
        lnl = np.full(chunk, np.nan)
        chi2 = np.full(chunk, np.nan)
        if hi > lo:
            a, b = self.evaluate(rows[lo:hi])
            lnl[: hi - lo] = a
            chi2[: hi - lo] = b
        if self.gather == "rccl":
            g_l, g_c = self._gather_rccl(lnl, chi2, chunk)
        else:
            g_l = self.dist.allgather_host(lnl, chunk)
            g_c = self.dist.allgather_host(chi2, chunk)
        return unpad(g_l, n, world), unpad(g_c, n, world)

    def _gather_rccl(self, lnl, chi2, chunk):
        eng = self.engine
        world = self.dist.world
        d_send = eng.alloc(2 * chunk)
        d_recv = eng.alloc(2 * chunk * world)
        try:
            eng.upload(d_send, np.concatenate([lnl, chi2]))
            eng.comm_allgather_async(d_send, d_recv, 2 * chunk)
            eng.sync()
            out = eng.download(d_recv, 2 * chunk * world).reshape(world, 2, chunk)
        finally:
            eng.free(d_send)
            eng.free(d_recv)
        return out[:, 0].reshape(-1), out[:, 1].reshape(-1)


class RcclGather:
    """All-gather of equal-size double arrays through RCCL on the engine's stream (device staging buffers are
    allocated once).  ``uid`` comes from rank 0's ``Engine.comm_unique_id()`` broadcast over the host process group.
    ``count`` doubles per rank and call - for :class:`victor_amd.sampler.DistributedEnsemble` a whole block of steps
    (``gather_block * walkers``): one upload, one ``ncclAllGather``, one download per block."""

    @classmethod
    def own_context(cls, fit, dist, count, device=None):
        """A gather with a GPU context (stream) of its OWN, created from ``fit``'s tables: an all-gather enqueued on a stream
        the walkers launch on would hold their launches back until the slowest rank has arrived
        (:class:`victor_amd.sampler.DistributedEnsemble` enqueues a block's exchange and collects it one block later)."""
        from .engine import Engine
        first = fit._get_engine()
        key = fit._engine_key(fit._merged({}))
        eng = Engine(fit, fit, device=first.device if device is None else device, matter_model=key, simpson_even=first.simpson_even,
                     lib=getattr(fit, "_native_lib", None))
        self = cls(eng, dist, count)
        self._own_engine = eng
        return self

    def __init__(self, engine, dist, count):
        self.engine, self.dist, self.count = engine, dist, int(count)
        if not one_device_per_rank(dist, engine):
            raise RuntimeError("two ranks share a GPU: RCCL needs one device per rank")
        uid = engine.comm_unique_id() if dist.rank == 0 else None
        uid = dist.broadcast_bytes(uid, src=0, nbytes=128)
        engine.comm_init(uid, dist.rank, dist.world)
        self.d_send = engine.alloc(self.count)
        self.d_recv = engine.alloc(self.count * dist.world)
        self.calls = 0

    def __call__(self, local):
        local = np.ascontiguousarray(local, dtype=np.float64)
        assert local.shape == (self.count,)
        self.engine.upload(self.d_send, local)
        self.engine.comm_allgather_async(self.d_send, self.d_recv, self.count)
        self.engine.sync()
        self.calls += 1
        return self.engine.download(self.d_recv, self.count * self.dist.world)

    # the same gather in two halves (vk_comm_allgather_host_begin / _finish): enqueue now, collect a block of steps later
    def begin(self, local):
        local = np.ascontiguousarray(local, dtype=np.float64)
        assert local.shape == (self.count,)
        self.engine.comm_allgather_host_begin(local)
        self.calls += 1

    def finish(self):
        return self.engine.comm_allgather_host_finish(self.count, self.dist.world)

    def close(self):
        self.engine.free(self.d_send)
        self.engine.free(self.d_recv)
        self.engine.comm_destroy()
        if getattr(self, "_own_engine", None) is not None:
            self._own_engine.close()
            self._own_engine = None


class DeviceGather:
    """All-gather of result vectors that stay in HBM, for the engines this process drives - the collective of a sharded batch.

    ``launched`` (one process per GPU under torchrun / mpirun / srun): ``ncclCommInitRank`` on ``engines[0]`` with the id rank 0
    draws, ``ncclAllGather`` on that context's stream.  Otherwise (ONE process, one context per GPU): ``ncclCommInitAll`` and a
    grouped all-gather.  ``mode`` says what was built: ``"rank"`` / ``"group"``, ``"host"`` when RCCL is missing or refuses
    (ranks sharing a device: a rehearsal on a one-GPU box) - the vectors then travel through the ranks' socket group and are
    written back into every receive buffer, a degraded mode kept so that the callers' checks read the same buffers -, ``"none"``
    for a single GPU without a launcher.  Every decision is taken collectively: all ranks end up in the same mode.
    A rendezvous that never completes raises :class:`victor_amd._native.CommInitTimeout`, which is fatal for the process."""

    NAMES = {"rank": "rccl allgather of lnL (ncclCommInitRank, one process per GPU)",
             "group": "rccl allgather of lnL (ncclCommInitAll, grouped calls, one process)",
             "host": "host allgather of lnL (RCCL unavailable or refused)",
             "none": "none (single GPU)"}

    def __init__(self, dist, engines, launched, log=None):
        from . import _native
        from .engine import Engine
        self.dist, self.engines, self.launched = dist, list(engines), bool(launched)
        self.mode = "none"
        log = log or (lambda msg: None)
        if not launched and len(self.engines) < 2:
            return
        ok = 1.0
        try:
            if launched:
                eng = self.engines[0]
                if not one_device_per_rank(dist, eng):
                    raise RuntimeError("two ranks share a GPU: RCCL needs one device per rank")
                try:
                    uid = eng.comm_unique_id() if dist.rank == 0 else None
                except Exception as exc:                # librccl missing: every rank must learn about it
                    log(f"rank {dist.rank}: RCCL unavailable ({exc})")
                    uid, ok = bytes(128), 0.0
                uid = dist.broadcast_bytes(uid, src=0, nbytes=128)
                ok = dist.min_float(ok)
                if ok:
                    try:
                        eng.comm_init(uid, dist.rank, dist.world)
                    except _native.CommInitTimeout:
                        raise
                    except Exception as exc:
                        log(f"rank {dist.rank}: ncclCommInitRank failed ({exc})")
                        ok = 0.0
                    ok = dist.min_float(ok)              # no collective before every rank holds a communicator
                    self._have_comm = True
            else:
                Engine.comm_init_all(self.engines)
                self._have_comm = True
        except _native.CommInitTimeout:
            raise
        except Exception as exc:
            log(f"rank {dist.rank}: RCCL communicator failed ({exc})")
            ok = 0.0
        ok = dist.min_float(ok)
        self.mode = ("rank" if launched else "group") if ok else "host"

    def degrade(self, failed):
        """After the first collective: every rank reports whether it failed; one failure sends all of them to the host mode."""
        if self.mode in ("rank", "group") and self.dist.min_float(0.0 if failed else 1.0) == 0.0:
            self.mode = "host"

    def __call__(self, d_send, d_recv, n):
        """Enqueue the all-gather of ``n`` doubles per GPU: ``d_send[i]`` / ``d_recv[i]`` live on ``engines[i]``'s device
        (``d_recv``: total GPUs x n doubles, rank-major).  ``"host"`` mode synchronises."""
        from .engine import Engine
        if self.mode == "rank":
            self.engines[0].comm_allgather_async(d_send[0], d_recv[0], n)
        elif self.mode == "group":
            Engine.comm_allgather_group_async(self.engines, list(d_send), list(d_recv), n)
        elif self.mode == "host":
            for e in self.engines:
                e.sync()
            local = np.concatenate([e.download(p, n) for e, p in zip(self.engines, d_send)])
            full = self.dist.allgather_host(local, len(local)) if self.launched else local
            for e, p in zip(self.engines, d_recv):
                e.upload(p, full)

    def rank_records(self):
        """Per GPU of the run, in rank order: what RCCL itself reports for its communicator (``Engine.comm_rank_info``) beside the
        PCI bus id the ranks compared before building it - [{"bus_id", "count", "rank", "device"}, ...]; ``count`` etc. are
        ``None`` in the host mode (no communicator).  Collective when launched."""
        import json
        mine = []
        for e in self.engines:
            info = e.comm_rank_info() if self.mode in ("rank", "group") else None
            mine.append(dict({"bus_id": e.bus_id()}, **(info or {"count": None, "rank": None, "device": None})))
        if not self.launched:
            return mine
        blob = json.dumps(mine).encode().ljust(512, b" ")[:512]
        out = []
        for r in range(self.dist.world):
            out.extend(json.loads(self.dist.broadcast_bytes(blob if r == self.dist.rank else None, src=r, nbytes=512).decode()))
        return out

    def close(self):
        if getattr(self, "_have_comm", False):
            for e in self.engines:
                try:
                    e.comm_destroy()
                except Exception:
                    pass
            self._have_comm = False


class MultiGPUFit:
    """One process driving several GPUs ("one host process per GPU, or one process driving G
    contexts"): a ``CCFFit`` per device, contiguous shards of every batch evaluated concurrently from host threads
    (ctypes releases the GIL for the duration of each library call; every context has its own stream).

    ``devices`` defaults to all visible GPUs.  The same device may be listed more than once (rehearsal on a one-GPU box).
    """

    def __init__(self, model, data, devices=None):
        from . import _native
        from .ccf_fit import CCFFit
        if devices is None:
            devices = list(range(max(_native.load().vk_device_count(), 1)))
        if not devices:
            raise ValueError("no devices given")
        self.devices = list(devices)
        self.fits = [CCFFit(model, data, device=d) for d in self.devices]
        self._pool = None

    def _executor(self):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=len(self.fits))
        return self._pool

    def _rows(self, params, kwargs):
        fit = self.fits[0]
        model = fit._merged(kwargs)
        return fit._fit_rows(params, model)

    def log_likelihood_batch(self, params, **kwargs):
        """(lnL[n], chi2[n]); same arguments as ``CCFFit.log_likelihood_batch``."""
        rows = self._rows(params, kwargs)
        n, g = len(rows), len(self.fits)
        bounds = [shard_bounds(n, g, r) for r in range(g)]
        jobs = [self._executor().submit(f.log_likelihood_batch, rows[lo:hi], **kwargs)
                for f, (lo, hi) in zip(self.fits, bounds) if hi > lo]
        parts = [j.result() for j in jobs]
        if not parts:
            return np.empty(0), np.empty(0)
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def enable_rccl(self):
        """Build RCCL communicators for the GPUs of this fit (``vk_comm_init_all``: one process, no rendezvous).  Returns
        False - and :meth:`log_likelihood_gathered` then goes through the host - when RCCL is missing or refuses (two
        contexts on one device)."""
        from .engine import Engine
        engines = [f._get_engine() for f in self.fits]
        try:
            Engine.comm_init_all(engines)
            self._rccl = True
        except Exception as exc:
            self._rccl_error = str(exc)
            self._rccl = False
        return self._rccl

    def log_likelihood_gathered(self, params, **kwargs):
        """lnL[n] of a batch sharded over the GPUs with the result all-gathered ON the GPUs (every device ends up holding the
        whole vector, as the ranks of a multi-process run do) and read back from the first one: parameter shards uploaded,
        one launch per device, one grouped RCCL all-gather over xGMI, one download.  Falls back to the host-side
        concatenation of :meth:`log_likelihood_batch` without communicators."""
        if not getattr(self, "_rccl", False):
            return self.log_likelihood_batch(params, **kwargs)[0]
        from . import _native as N
        from .engine import Engine
        rows = self._rows(params, kwargs)
        n, g = len(rows), len(self.fits)
        if n == 0:
            return np.empty(0)
        chunk = padded_chunk(n, g)
        fit0 = self.fits[0]
        model = fit0._merged(kwargs)
        fit_options = fit0._merged_fit(kwargs)
        engines = [f._get_engine(f._engine_key(model), model["simpson_even"]) for f in self.fits]
        if engines != [f._get_engine() for f in self.fits]:
            return self.log_likelihood_batch(params, **kwargs)[0]      # another table set than the one the communicators belong to
        bufs = getattr(self, "_gather_bufs", None)
        if bufs is None or bufs[0] < chunk:
            if bufs is not None:
                for e, ptrs in zip(engines, bufs[1]):
                    for p in ptrs:
                        e.free(p)
            cap = max(chunk, 64)
            bufs = self._gather_bufs = (cap, [(e.alloc(cap * N.VK_NPAR), e.alloc(cap), e.alloc(cap), e.alloc(cap * e.n_data),
                                               e.alloc(cap * g)) for e in engines])
        for r, (f, e, (d_rows, d_lnl, d_chi, d_ws, d_all)) in enumerate(zip(self.fits, engines, bufs[1])):
            lo, hi = shard_bounds(n, g, r)
            shard = np.empty((chunk, N.VK_NPAR))
            shard[: hi - lo] = rows[lo:hi]
            shard[hi - lo:] = rows[hi - 1 if hi > lo else 0]          # padding rows: valid parameters, results dropped by unpad
            e.upload(d_rows, shard)
            e.eval_device_async(e.make_opts(model, fit_options), d_rows, chunk, d_lnl, d_chi, d_ws)
        Engine.comm_allgather_group_async(engines, [b[1] for b in bufs[1]], [b[4] for b in bufs[1]], chunk)
        for e in engines:
            e.sync()
        return unpad(engines[0].download(bufs[1][0][4], chunk * g), n, g)

    def theory_vector_batch(self, params, **kwargs):
        rows = self._rows(params, kwargs)
        n, g = len(rows), len(self.fits)
        bounds = [shard_bounds(n, g, r) for r in range(g)]
        jobs = [self._executor().submit(f.theory_vector_batch, rows[lo:hi], **kwargs)
                for f, (lo, hi) in zip(self.fits, bounds) if hi > lo]
        parts = [j.result() for j in jobs]
        return np.concatenate(parts) if parts else np.empty((0, 0))

    def close(self):
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        if getattr(self, "_rccl", False):
            for f in self.fits:
                f._get_engine().comm_destroy()
            self._rccl = False
