"""Multi-GPU path on CPU: batch sharding + gather with world_size 2 over gloo (no GPU needed).

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
dist.init_process_group("gloo")
assert dist.world == 2
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
dist.barrier()
print("rank", dist.rank, "ok")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, OMP_NUM_THREADS="1")
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
dist.init_process_group("gloo")
specs = [ParamSpec("a", -1, 1, 0.2, 0.05, 0.1), ParamSpec("b", -4, 2, -1.0, 0.2, 0.5)]
ens = DistributedEnsemble(evaluate, specs, walkers_per_rank=8, dist=dist, seed=5)
chain, lnl, all_lnl = ens.run(25)
assert chain.shape == (25, 8, 2) and all_lnl.shape == (25, 16)
# every rank sees the whole ensemble's log-likelihoods, its own slice in its own slot
assert np.array_equal(all_lnl[:, dist.rank * 8:(dist.rank + 1) * 8], lnl)
other = all_lnl[:, (1 - dist.rank) * 8:(2 - dist.rank) * 8]
assert not np.array_equal(other, lnl)          # the ranks run different walkers (different seeds)
print("rank", dist.rank, "ok", float(all_lnl.sum()))
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
