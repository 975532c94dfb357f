"""Host-side rendezvous of the ranks of a multi-process run, on nothing but the standard library.

One process per GPU needs three things from its peers before and around the data path: the 128-byte RCCL unique id of
rank 0, barriers, and a few scalars (max / min of a timing, every rank's kernel time).  The log-likelihoods themselves travel
over RCCL (``vk_comm_allgather_async``); when no communicator can be built - ranks sharing one GPU in a rehearsal, librccl
missing - they fall back to :meth:`SocketGroup.allgather_doubles` here.

Topology: a star.  Rank 0 listens, every other rank connects once and keeps its connection; every collective is one
length-prefixed message from each rank to rank 0 and one reply.  A message costs ~50 us on loopback, which is nothing next
to the steps it brackets (a barrier per timed region, not per step).

Where the ranks meet (first match):

``VICTOR_RDZV=host:port``
    explicit TCP endpoint (multi-node runs, e.g. cobaya under ``mpirun``: export it in the job script).
``VICTOR_RDZV=unix:/path``
    explicit Unix-domain socket.
launcher environment of a single-node job - ``MASTER_ADDR`` a literal loopback address (``torch.distributed.run --master-addr
127.0.0.1``, the bench driver), or the launcher's local world size equal to the world size
    a Unix-domain socket in the temp directory named after ``MASTER_PORT`` (and ``TORCHELASTIC_RUN_ID``): the launcher's own
    store already listens on ``MASTER_PORT`` itself, so that port is not ours to bind.
any other launcher environment (several nodes)
    TCP on ``MASTER_PORT + 1`` of ``MASTER_ADDR``; rank 0 binds every interface.
The choice depends on nothing but the job's environment - never on the name of the host a rank runs on - so every rank of
a job makes the same one.

Rank and world size come from ``RANK`` / ``WORLD_SIZE`` (torchrun), ``OMPI_COMM_WORLD_RANK`` / ``_SIZE`` (Open MPI),
``PMI_RANK`` / ``PMI_SIZE`` (MPICH, Intel MPI) or - inside an ``srun`` step only: ``sbatch`` exports the same variables into
the batch shell, where a plain ``python bench.py`` is one process, not rank 0 of many - ``SLURM_PROCID`` / ``SLURM_NTASKS``.

The reference has no counterpart: its chains are independent processes under ``mpirun`` (README.md:30).
"""

import os
import socket
import struct
import tempfile
import time

_MAGIC = b"VKRZ1\0\0\0"
_RANK_KEYS = (("RANK", "WORLD_SIZE"), ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE"), ("PMI_RANK", "PMI_SIZE"),
              ("SLURM_PROCID", "SLURM_NTASKS"))
_LOCAL_KEYS = ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "MV2_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID")


def launcher_ranks(environ=None):
    """(rank, world, local_rank) from the launcher's environment, or ``None`` when the process was started on its own."""
    env = os.environ if environ is None else environ
    in_srun_step = env.get("SLURM_STEP_ID", "") != "" or env.get("SLURM_SRUN_COMM_HOST", "") != ""
    for rk, wk in _RANK_KEYS:
        if rk.startswith("SLURM_") and not in_srun_step:
            continue                                   # the batch shell of an sbatch job, not a launched task
        if env.get(rk, "") != "" and env.get(wk, "") != "":
            rank, world = int(env[rk]), int(env[wk])
            local = rank
            for lk in _LOCAL_KEYS:
                if env.get(lk, "") != "":
                    local = int(env[lk])
                    break
            return rank, world, local
    return None


def endpoint(environ=None):
    """Where the ranks meet: ``("unix", path)`` or ``("tcp", (host, port))`` - see the module docstring."""
    env = os.environ if environ is None else environ
    spec = env.get("VICTOR_RDZV", "")
    if spec.startswith("unix:"):
        return "unix", spec[5:]
    if spec:
        host, _, port = spec.rpartition(":")
        return "tcp", (host or "127.0.0.1", int(port))
    addr = env.get("MASTER_ADDR", "127.0.0.1")
    port = int(env.get("MASTER_PORT", "29400"))
    found = launcher_ranks(env)
    world = found[1] if found else 1
    local_world = 0
    for key in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "MV2_COMM_WORLD_LOCAL_SIZE"):
        if env.get(key, "") != "":
            local_world = int(env[key])
            break
    single_node = addr in ("127.0.0.1", "localhost", "::1") or (local_world > 0 and local_world == world)
    if single_node:
        run = "".join(c for c in env.get("TORCHELASTIC_RUN_ID", "") if c.isalnum())[:24]
        name = f"victor_rdzv_{os.getuid()}_{port}{'_' + run if run else ''}.sock"
        return "unix", os.path.join(tempfile.gettempdir(), name)
    return "tcp", (addr, port + 1)


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += part
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class SocketGroup:
    """The ranks of one job.  Every method is a collective: all ranks must call it, in the same order."""

    def __init__(self, rank, world, where=None, timeout=120.0, collective_timeout=None):
        """``timeout`` bounds the handshake (rank 0 waiting for its peers, a peer waiting for rank 0).  The collectives after
        it wait ``collective_timeout`` seconds for a message (default ``None``: as long as it takes - ranks may be minutes
        apart, e.g. while one of them compiles the library)."""
        self.rank, self.world = int(rank), int(world)
        if not (0 <= self.rank < self.world):
            raise ValueError("bad rank/world")
        self._peers = []          # rank 0: sockets of ranks 1..world-1, in rank order
        self._root = None         # other ranks: socket to rank 0
        self._path = None
        if self.world == 1:
            return
        kind, addr = where or endpoint()
        family = socket.AF_UNIX if kind == "unix" else socket.AF_INET
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            if kind == "unix":
                try:
                    os.unlink(addr)                      # a stale socket file of a finished job
                except OSError:
                    pass
                self._path = addr
            srv = socket.socket(family, socket.SOCK_STREAM)
            if kind == "tcp":
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("", addr[1]) if kind == "tcp" else addr)        # TCP: every interface (MASTER_ADDR names this node somehow)
            srv.listen(self.world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < self.world - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                hello = _recv(conn)
                if len(hello) != 16 or hello[:8] != _MAGIC:
                    conn.close()                          # not one of ours
                    continue
                peer, peer_world = struct.unpack("<ii", hello[8:])
                if peer_world != self.world or not (0 < peer < self.world) or peer in by_rank:
                    conn.close()
                    raise RuntimeError(f"rendezvous: unexpected peer (rank {peer} of {peer_world})")
                if kind == "tcp":
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                by_rank[peer] = conn
            srv.close()
            self._peers = [by_rank[r] for r in range(1, self.world)]
            for conn in self._peers:
                _send(conn, _MAGIC)                       # everyone is here
                conn.settimeout(collective_timeout)
        else:
            while True:
                sock = socket.socket(family, socket.SOCK_STREAM)
                try:
                    sock.connect(addr)
                    break
                except (ConnectionRefusedError, FileNotFoundError, socket.timeout):
                    sock.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 did not appear at {addr}")
                    time.sleep(0.02)
            sock.settimeout(timeout)
            if kind == "tcp":
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(sock, _MAGIC + struct.pack("<ii", self.rank, self.world))
            if _recv(sock) != _MAGIC:
                raise RuntimeError("rendezvous: bad reply from rank 0")
            sock.settimeout(collective_timeout)
            self._root = sock

    # ---- the one primitive: everybody's bytes to everybody --------------------------------------------------------
    def allgather_bytes(self, payload, what="allgather"):
        """``[payload of rank 0, payload of rank 1, ...]`` on every rank.  Every message carries the call's sequence number and
        the name of the collective: ranks that are out of step (one took a branch the others did not) get a RuntimeError
        naming both calls instead of each other's bytes."""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        self._seq = getattr(self, "_seq", 0) + 1
        tag = struct.pack("<I", self._seq) + what.encode()[:12].ljust(12, b"\0")
        if self.rank == 0:
            got = [_recv(conn) for conn in self._peers]
            bad = [(r + 1, g[:16]) for r, g in enumerate(got) if g[:16] != tag]
            parts = [payload] + [g[16:] for g in got]
            blob = (b"\x01" + repr(bad).encode()) if bad else b"\x00" + b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for conn in self._peers:
                _send(conn, blob)
            if bad:
                raise RuntimeError(f"rendezvous: ranks out of step - rank 0 is in call {self._seq} ({what}), got {bad}")
            return parts
        _send(self._root, tag + payload)
        blob = _recv(self._root)
        if blob[:1] != b"\x00":
            raise RuntimeError(f"rendezvous: ranks out of step at call {self._seq} ({what}) of rank {self.rank}: {blob[1:200].decode(errors='replace')}")
        parts, pos = [], 1
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, pos)
            parts.append(blob[pos + 8: pos + 8 + n])
            pos += 8 + n
        return parts

    def barrier(self):
        self.allgather_bytes(b"", "barrier")

    def broadcast_bytes(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src and payload is not None else b"", "broadcast")[src]

    def allgather_doubles(self, values):
        """Concatenation, in rank order, of every rank's array of doubles (the arrays may differ in length)."""
        import numpy as np
        local = np.ascontiguousarray(values, dtype=np.float64)
        parts = self.allgather_bytes(local.tobytes(), "doubles")
        return np.concatenate([np.frombuffer(p, dtype=np.float64) for p in parts]) if parts else np.empty(0)

    def max_float(self, x):
        return max(struct.unpack("<d", p)[0] for p in self.allgather_bytes(struct.pack("<d", float(x)), "max"))

    def min_float(self, x):
        return min(struct.unpack("<d", p)[0] for p in self.allgather_bytes(struct.pack("<d", float(x)), "min"))

    def close(self):
        for conn in self._peers:
            conn.close()
        self._peers = []
        if self._root is not None:
            self._root.close()
            self._root = None
        if self._path:
            try:
                os.unlink(self._path)
            except OSError:
                pass
            self._path = None
