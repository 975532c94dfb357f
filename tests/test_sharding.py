"""Multi-GPU path on CPU: batch sharding + gather with world_size 2 over the standard-library socket group (no GPU needed).

The device evaluation is replaced by a deterministic stand-in (the sharding layer only needs
``rows -> (lnl, chi2)``); what is under test is the partitioning, the padding to equal counts, the gather
and the reassembly in batch order - the parts that differ between 1 and N ranks.
"""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from victor_amd.sharding import padded_chunk, shard_bounds, unpad

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [0, 1, 2, 7, 8, 9, 1024, 65536, 65537])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_shard_bounds_partition(n, world):
    edges = [shard_bounds(n, world, r) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n
    for (a, b), (c, d) in zip(edges[:-1], edges[1:]):
        assert b == c and b >= a
    sizes = [b - a for a, b in edges]
    assert max(sizes) - min(sizes) <= 1 and max(sizes) == padded_chunk(n, world) or n == 0
    with pytest.raises(ValueError):
        shard_bounds(n, world, world)


def test_unpad_restores_batch_order():
    n, world = 11, 4
    chunk = padded_chunk(n, world)
    gathered = np.full(world * chunk, np.nan)
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        gathered[r * chunk: r * chunk + hi - lo] = np.arange(lo, hi)
    assert np.array_equal(unpad(gathered, n, world), np.arange(n))


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from victor_amd.sharding import Dist, ShardedLikelihood

def evaluate(rows):                       # stand-in for CCFFit.log_likelihood_batch on this rank's GPU
    chi2 = (rows ** 2).sum(axis=1)
    return -0.5 * chi2, chi2

dist = Dist()
dist.connect()
assert dist.world == 2
# a single-node job: RCCL's bootstrap sockets go over the loopback interface unless the user has chosen one (Dist.connect)
assert os.environ["NCCL_SOCKET_IFNAME"] == os.environ.get("EXPECT_IFNAME", "lo")
rng = np.random.default_rng(5)
out = []
for n in (1, 2, 9, 1000, 1001):
    rows = rng.normal(size=(n, 10))
    lnl, chi2 = ShardedLikelihood(evaluate, dist, gather="host")(rows)
    want_l, want_c = evaluate(rows)
    assert lnl.shape == (n,) and np.array_equal(lnl, want_l) and np.array_equal(chi2, want_c), n
t = dist.max_float(1.0 + dist.rank)
assert dist.min_float(1.0 + dist.rank) == 1.0
assert t == 2.0
payload = dist.broadcast_bytes(bytes(range(128)) if dist.rank == 0 else None, src=0, nbytes=128)
assert payload == bytes(range(128))
got = dist.group.allgather_doubles(np.arange(3 + dist.rank, dtype=float))       # unequal lengths
assert np.array_equal(got, np.array([0., 1., 2., 0., 1., 2., 3.]))
dist.barrier()
dist.close()
print("rank", dist.rank, "ok")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_socket_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, OMP_NUM_THREADS="1")
        env.pop("NCCL_SOCKET_IFNAME", None)
        if rank == 1:                               # a user's own choice is left alone
            env.update(NCCL_SOCKET_IFNAME="eth7", EXPECT_IFNAME="eth7")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out


ENSEMBLE_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from victor_amd.sharding import Dist
from victor_amd.sampler import DistributedEnsemble, ParamSpec

def evaluate(batch):
    return -0.5 * ((batch["a"] - 0.2) ** 2 / 0.01 + (batch["b"] + 1.0) ** 2 / 0.25)

dist = Dist()
dist.connect()
specs = [ParamSpec("a", -1, 1, 0.2, 0.05, 0.1), ParamSpec("b", -4, 2, -1.0, 0.2, 0.5)]
# the steps' log-likelihoods are exchanged in blocks of `gather_block` steps (default 64, the random-number block): the
# chain and the gathered history must not depend on the block length, and 150 steps are ceil(150 / K) collectives
runs = {{}}
for K in (1, 7, 64, None):
    ens = DistributedEnsemble(evaluate, specs, walkers_per_rank=8, dist=dist, seed=5, gather_block=K)
    chain, lnl, all_lnl = ens.run(150)
    K = ens.gather_block
    assert chain.shape == (150, 8, 2) and all_lnl.shape == (150, 16) and np.all(np.isfinite(all_lnl))
    assert ens.n_collectives == -(-150 // K), (K, ens.n_collectives)
    # every rank sees the whole ensemble's log-likelihoods, its own slice in its own slot
    assert np.array_equal(all_lnl[:, dist.rank * 8:(dist.rank + 1) * 8], lnl)
    other = all_lnl[:, (1 - dist.rank) * 8:(2 - dist.rank) * 8]
    assert not np.array_equal(other, lnl)          # the ranks run different walkers (different seeds)
    runs[K] = (chain, all_lnl)
assert sorted(runs) == [1, 7, 64]              # None = the sampler's 64-step block
for K in (7, 64):
    assert np.array_equal(runs[K][0], runs[1][0]) and np.array_equal(runs[K][1], runs[1][1]), K
# a second run() continues the history (two more collectives at K = 64: 100 steps)
more = ens.run(100)[2]
assert more.shape == (250, 16) and ens.n_collectives == 3 + 2 and np.array_equal(more[:150], runs[64][1])
# a gather that comes in two halves (RcclGather.begin / finish) is enqueued behind a block and collected one block later:
# the same history, the same number of collectives, at most one exchange in flight
class TwoHalves:
    def __init__(self):
        self.sent, self.in_flight, self.max_in_flight = None, 0, 0
    def begin(self, local):
        assert self.sent is None
        self.sent = np.array(local)
        self.in_flight += 1
        self.max_in_flight = max(self.max_in_flight, self.in_flight)
    def finish(self):
        local, self.sent = self.sent, None
        self.in_flight -= 1
        return dist.allgather_host(local, len(local))
for K in (7, 64):
    g = TwoHalves()
    late = DistributedEnsemble(evaluate, specs, walkers_per_rank=8, dist=dist, seed=5, gather=g, gather_block=K)
    assert late.overlap
    chain, lnl, all_lnl = late.run(150)
    assert np.array_equal(chain, runs[K][0]) and np.array_equal(all_lnl, runs[K][1]) and late.n_collectives == -(-150 // K)
    assert g.sent is None and g.in_flight == 0 and g.max_in_flight == 1
print("rank", dist.rank, "ok", float(runs[1][1].sum()))
'''


def test_two_rank_distributed_ensemble(tmp_path):
    script = tmp_path / "ens_worker.py"
    script.write_text(ENSEMBLE_WORKER.format(root=ROOT))
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    sums = []
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out
        sums.append(out.strip().split()[-1])
    assert sums[0] == sums[1]                       # both ranks hold the identical gathered log-likelihood history


def test_default_device_follows_the_launcher():
    from victor_amd.sharding import default_device
    assert default_device({}, 8) == 0
    assert default_device({"VICTOR_HIP_DEVICE": "5", "LOCAL_RANK": "2"}, 8) == 5
    assert default_device({"OMPI_COMM_WORLD_LOCAL_RANK": "3"}, 8) == 3
    assert default_device({"SLURM_LOCALID": "9"}, 8) == 1
    assert default_device({"LOCAL_RANK": "2"}, 1) == 0


def test_rendezvous_endpoint_and_launcher_environment():
    """Where the ranks meet and who they are, for every launcher the plug-in is run under (rendezvous.py docstring)."""
    from victor_amd import rendezvous as rz
    assert rz.launcher_ranks({}) is None
    assert rz.launcher_ranks({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "1"}) == (3, 8, 1)
    assert rz.launcher_ranks({"OMPI_COMM_WORLD_RANK": "2", "OMPI_COMM_WORLD_SIZE": "4", "OMPI_COMM_WORLD_LOCAL_RANK": "0"}) == (2, 4, 0)
    assert rz.launcher_ranks({"PMI_RANK": "1", "PMI_SIZE": "2"}) == (1, 2, 1)
    # Slurm: only inside an srun step - sbatch exports SLURM_PROCID / SLURM_NTASKS into the batch shell as well, where a plain
    # `python bench.py` is ONE process and must not wait for 15 peers that will never come
    assert rz.launcher_ranks({"SLURM_PROCID": "0", "SLURM_NTASKS": "16", "SLURM_LOCALID": "0"}) is None
    assert rz.launcher_ranks({"SLURM_PROCID": "5", "SLURM_NTASKS": "16", "SLURM_LOCALID": "5", "SLURM_STEP_ID": "0"}) == (5, 16, 5)
    assert rz.launcher_ranks({"SLURM_PROCID": "5", "SLURM_NTASKS": "16", "SLURM_SRUN_COMM_HOST": "10.0.0.1"}) == (5, 16, 5)
    assert rz.endpoint({"VICTOR_RDZV": "node17:4711"}) == ("tcp", ("node17", 4711))
    assert rz.endpoint({"VICTOR_RDZV": "unix:/tmp/x.sock"}) == ("unix", "/tmp/x.sock")
    kind, path = rz.endpoint({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511", "TORCHELASTIC_RUN_ID": "none"})
    assert kind == "unix" and "29511" in path and path.endswith(".sock")      # the launcher's own store owns the TCP port
    assert rz.endpoint({"MASTER_ADDR": "10.1.2.3", "MASTER_PORT": "29511"}) == ("tcp", ("10.1.2.3", 29512))


def test_every_rank_of_a_job_chooses_the_same_endpoint(monkeypatch):
    """The transport must follow from the job's environment alone: with MASTER_ADDR = the first node's host name the ranks ON
    that node used to pick the Unix socket and the ranks on the other nodes TCP, where nobody listened."""
    import socket
    from victor_amd import rendezvous as rz
    env = {"MASTER_ADDR": "node0", "MASTER_PORT": "29400", "RANK": "0", "WORLD_SIZE": "16", "LOCAL_WORLD_SIZE": "8"}
    seen = set()
    for host in ("node0", "node1"):
        monkeypatch.setattr(socket, "gethostname", lambda h=host: h)
        seen.add(rz.endpoint(env))
    assert seen == {("tcp", ("node0", 29401))}
    # a single-node job may name its node any way it likes: local world size == world size -> the Unix socket, on every rank
    one = dict(env, WORLD_SIZE="8")
    seen = set()
    for host in ("node0", "node1"):
        monkeypatch.setattr(socket, "gethostname", lambda h=host: h)
        seen.add(rz.endpoint(one)[0])
    assert seen == {"unix"}
    assert rz.endpoint({"MASTER_ADDR": "localhost", "MASTER_PORT": "1234"})[0] == "unix"


def test_collectives_outlive_the_handshake_timeout(tmp_path):
    """The handshake timeout must not govern the collectives after it: ranks may be far apart (one of them compiling the
    library) - here rank 1 arrives at the barrier later than the whole handshake timeout."""
    import multiprocessing as mp
    where = ("unix", str(tmp_path / "late.sock"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_late_rank, args=(r, where, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert got == [(0, "ok"), (1, "ok")]


def _late_rank(rank, where, q):
    import time
    from victor_amd.rendezvous import SocketGroup
    try:
        grp = SocketGroup(rank, 2, where=where, timeout=1.0)
        if rank == 1:
            time.sleep(2.5)
        grp.barrier()
        grp.close()
        q.put((rank, "ok"))
    except Exception as exc:       # noqa: BLE001 - reported to the parent
        q.put((rank, repr(exc)))


def test_socket_group_collectives_with_three_ranks(tmp_path):
    """Three ranks as threads of one process over a Unix socket and over TCP: all-gather of bytes, broadcast, barrier,
    max / min, doubles of unequal lengths; a stray connection that does not speak the protocol is ignored."""
    import threading
    from victor_amd.rendezvous import SocketGroup
    for where in (("unix", str(tmp_path / "g.sock")), ("tcp", ("127.0.0.1", _free_port()))):
        results, errors = {}, []

        def run(rank):
            try:
                g = SocketGroup(rank, 3, where=where, timeout=30)
                parts = g.allgather_bytes(bytes([rank]) * (rank + 1))
                uid = g.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
                g.barrier()
                results[rank] = (parts, uid, g.max_float(rank), g.min_float(rank), g.allgather_doubles(np.full(rank, rank)).tolist())
                g.close()
            except Exception as exc:            # pragma: no cover
                errors.append((rank, repr(exc)))

        threads = [threading.Thread(target=run, args=(r,)) for r in range(3)]
        threads[0].start()
        if where[0] == "tcp":                   # a port scanner's connect: closed without the greeting
            import socket as so
            import time
            time.sleep(0.2)
            stray = so.create_connection(where[1])
            stray.sendall(b"\x03\x00\x00\x00\x00\x00\x00\x00abc")
            stray.close()
        for t in threads[1:]:
            t.start()
        for t in threads:
            t.join(60)
        assert not errors, errors
        for rank in range(3):
            parts, uid, mx, mn, dbl = results[rank]
            assert parts == [b"\x00", b"\x01\x01", b"\x02\x02\x02"] and uid == bytes(range(128))
            assert (mx, mn) == (2.0, 0.0) and dbl == [1.0, 2.0, 2.0]


def test_socket_group_reports_ranks_that_are_out_of_step(tmp_path):
    """One rank calls a barrier where the other calls max_float (a branch only one of them took): both get a RuntimeError
    naming the two calls, instead of hanging or reading each other's payload as their own."""
    import threading
    from victor_amd.rendezvous import SocketGroup
    where = ("unix", str(tmp_path / "oos.sock"))
    errors = {}

    def run(rank):
        g = SocketGroup(rank, 2, where=where, timeout=30)
        g.barrier()
        try:
            if rank == 0:
                g.barrier()
            else:
                g.max_float(1.0)
        except RuntimeError as exc:
            errors[rank] = str(exc)
        g.close()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert set(errors) == {0, 1} and all("out of step" in e for e in errors.values()), errors
    assert "barrier" in errors[0] and "max" in errors[0]


_EIGHT_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from victor_amd.sharding import Dist
dist = Dist()
assert dist.launched and dist.world == 8 and dist.rank == int(os.environ["RANK"]) and dist.local_rank == int(os.environ["LOCAL_RANK"])
dist.connect(timeout=60)
dist.barrier()
uid = dist.broadcast_bytes(bytes(range(128)) if dist.rank == 0 else None, src=0, nbytes=128)      # bench.py: the RCCL unique id
assert uid == bytes(range(128))
assert dist.min_float(1.0 if dist.rank != 5 else 0.0) == 0.0                                        # one rank without a communicator
B = 65536
mine = np.arange(dist.rank * B, (dist.rank + 1) * B, dtype=float)                                   # the host-gather fallback's payload
every = dist.allgather_host(mine, B)
assert every.shape == (8 * B,) and np.array_equal(every, np.arange(8 * B, dtype=float))
assert dist.max_float(10.0 + dist.rank) == 17.0
dist.barrier()
dist.close()
print("ok", dist.rank)
"""


def test_eight_launched_ranks_meet_and_gather(tmp_path):
    """The host side of `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`, without the GPUs: eight
    processes with the launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) find each other through
    Dist().connect() - the Unix socket derived from MASTER_PORT - and run bench.py's sequence of host collectives, including
    the 8 x 65536-double host gather of the degraded mode."""
    import subprocess
    import sys
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(_EIGHT_RANK_WORKER)
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TORCHELASTIC_RUN_ID=f"t{port}", TMPDIR=str(tmp_path))
        env.pop("VICTOR_RDZV", None)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=120) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and out.strip() == f"ok {r}", (r, err[-800:])
