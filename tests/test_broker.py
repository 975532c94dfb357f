"""Mailbox protocol of victor_amd/broker.py on the CPU: segment layout, attach / detach, the chain-side call against a stand-in
server thread that speaks the protocol of ``vk_serve_mailboxes`` (the native loop itself needs a GPU context: test_gpu_broker.py)."""

import ctypes as C
import os
import threading
import time
import uuid

import numpy as np
import pytest

from victor_amd import _native as N
from victor_amd import broker as B


def _name():
    return f"victor_test_{os.getpid()}_{uuid.uuid4().hex[:8]}"


class StandInServer(threading.Thread):
    """Answers every pending mailbox with lnl = sum(row), chi2 = row[0] * 2 (in Python; same words, same order as the native loop)."""

    def __init__(self, seg):
        super().__init__(daemon=True)
        self.seg = seg
        self.stop = False
        self.batches = []

    def run(self):
        boxes = self.seg.boxes
        while not self.stop:
            pending = [(i, b.req_seq) for i, b in enumerate(boxes) if b.state == B.N_BOX_ATTACHED and b.req_seq != b.resp_seq]
            if not pending:
                time.sleep(0.0002)
                continue
            self.batches.append(len(pending))
            for i, seq in pending:
                b = boxes[i]
                row = list(b.row)
                b.lnl, b.chi2, b.status = sum(row), 2.0 * row[0], 0
                b.resp_seq = seq


def _ready_segment(name, digest, n_slots=4):
    seg = B._Segment(B.shm_path(name), create=True, n_slots=n_slots)
    seg.header.digest = digest.encode()
    seg.header.server_pid = os.getpid()
    seg.header.state = B.READY
    return seg


def test_layout_matches_the_header():
    """Offsets the Python client packs against (include/victor_hip.h: vk_mailbox; the library static_asserts the same numbers)."""
    assert C.sizeof(N.vk_mailbox) == 256
    assert (N.vk_mailbox.req_seq.offset, N.vk_mailbox.state.offset, N.vk_mailbox.client_pid.offset) == (0, 8, 16)
    assert (N.vk_mailbox.row.offset, N.vk_mailbox.resp_seq.offset) == (64, 192)
    assert (N.vk_mailbox.lnl.offset, N.vk_mailbox.chi2.offset, N.vk_mailbox.status.offset) == (200, 208, 216)
    assert C.sizeof(B._Header) <= B.HEADER_BYTES


def test_client_round_trips_and_detaches():
    name, digest = _name(), B.config_digest({"a": 1}, {"b": [1, 2]})
    seg = _ready_segment(name, digest)
    srv = StandInServer(seg)
    srv.start()
    try:
        c1 = B.BrokerClient(name, digest, timeout=5)
        c2 = B.BrokerClient(name, digest, timeout=5)
        assert {c1.slot, c2.slot} == {0, 1}
        rng = np.random.default_rng(0)
        for _ in range(50):
            row = rng.normal(size=N.VK_NPAR).tolist()
            lnl, chi2 = c1.eval_point(row)
            assert lnl == sum(row) and chi2 == 2.0 * row[0]
            assert c2.eval_point(row[::-1])[1] == 2.0 * row[-1]
        assert seg.boxes[c1.slot].client_pid == os.getpid()
        slot = c1.slot
        c1.close()
        assert seg.boxes[slot].state == B.N_BOX_FREE
        c3 = B.BrokerClient(name, digest, timeout=5)          # the freed mailbox is handed out again, sequence words reset
        assert c3.slot == slot and c3.eval_point([1.0] * N.VK_NPAR) == (float(N.VK_NPAR), 2.0)
        c2.close()
        c3.close()
    finally:
        srv.stop = True
        srv.join(2)
        path = seg.path
        seg.close()
        for p in (path, path + ".lock"):
            if os.path.exists(p):
                os.unlink(p)


def test_client_refuses_another_configuration_and_reports_failures():
    name, digest = _name(), B.config_digest({"a": 1}, {})
    seg = _ready_segment(name, digest, n_slots=1)
    try:
        with pytest.raises(B.InputError, match="another"):
            B.BrokerClient(name, B.config_digest({"a": 2}, {}), timeout=2)
        c = B.BrokerClient(name, digest, timeout=2)
        with pytest.raises(N.NativeError, match="taken"):
            B.BrokerClient(name, digest, timeout=2)
        c.close()
        seg.header.error = b"no HIP device visible"
        seg.header.state = B.FAILED
        with pytest.raises(N.NativeError, match="no HIP device"):
            B.BrokerClient(name, digest, timeout=2)
    finally:
        path = seg.path
        seg.close()
        for p in (path, path + ".lock"):
            if os.path.exists(p):
                os.unlink(p)
    with pytest.raises(N.NativeError, match="no broker segment"):
        B.BrokerClient(_name(), digest, timeout=0.2)


def test_a_chain_does_not_wait_for_an_owner_that_died_while_starting():
    """A segment in state STARTING whose owner process has gone (crashed or was killed during HIP initialisation, before it could
    write FAILED): the chains of the job learn it at once instead of waiting out the five-minute time-out."""
    import subprocess
    import sys
    name, digest = _name(), "d" * 64
    seg = B._Segment(B.shm_path(name), create=True, n_slots=2)
    try:
        dead = subprocess.Popen([sys.executable, "-c", "pass"])
        dead.wait()
        seg.header.digest = digest.encode()
        seg.header.server_pid = dead.pid
        seg.header.state = B.STARTING
        t0 = time.monotonic()
        with pytest.raises(N.NativeError, match="died while starting"):
            B.BrokerClient(name, digest, timeout=300)
        assert time.monotonic() - t0 < 5
        # an owner that is alive and still starting is waited for (here: until the caller's own time-out)
        seg.header.server_pid = os.getpid()
        with pytest.raises(N.NativeError, match="did not become ready"):
            B.BrokerClient(name, digest, timeout=0.3)
    finally:
        path = seg.path
        seg.close()
        for p in (path, path + ".lock"):
            if os.path.exists(p):
                os.unlink(p)


def test_an_error_status_reaches_the_caller():
    name, digest = _name(), "d" * 64
    seg = _ready_segment(name, digest, n_slots=1)

    def answer_with_error():
        b = seg.boxes[0]
        while b.req_seq == b.resp_seq:
            time.sleep(0.0005)
        b.lnl, b.chi2, b.status = -np.inf, np.inf, -2
        b.resp_seq = b.req_seq

    t = threading.Thread(target=answer_with_error, daemon=True)
    try:
        c = B.BrokerClient(name, digest, timeout=2)
        t.start()
        with pytest.raises(N.NativeError, match="error -2"):
            c.eval_point([0.0] * N.VK_NPAR)
        c.close()
    finally:
        t.join(2)
        path = seg.path
        seg.close()
        for p in (path, path + ".lock"):
            if os.path.exists(p):
                os.unlink(p)


def test_digest_and_device_selection():
    m, d = {"z_eff": 0.57, "dir": "x"}, {"likelihood": {"form": "Sellentin", "nmocks": 1000}}
    assert B.config_digest(m, d) == B.config_digest(dict(reversed(list(m.items()))), d)       # key order does not matter
    assert B.config_digest(m, d) != B.config_digest(dict(m, z_eff=0.58), d)
    assert B.broker_device({}) == 0
    assert B.broker_device({"VICTOR_HIP_DEVICE": "3", "LOCAL_RANK": "1"}) == 3
    env = {"RANK": "5", "WORLD_SIZE": "16", "LOCAL_RANK": "5", "VICTOR_HIP_BROKER_GPUS": "4"}
    assert B.broker_device(env) == 1
    assert B.broker_device({"RANK": "5", "WORLD_SIZE": "16", "LOCAL_RANK": "5"}) == 0
    assert B.auto_name("ab" * 32, 2).endswith("_gpu2")
    with pytest.raises(B.InputError):
        B.shm_path("../etc/passwd")


def test_brokered_fit_never_builds_a_gpu_context(monkeypatch):
    """With a broker configured, CCFFit.log_likelihood goes to the mailbox and the process creates no Engine; keyword overrides
    and changed option dictionaries fall back to a context of their own (here: fail loudly, there is no GPU in this container)."""
    import victor_amd
    from tests import cases
    model, data = cases.boss_options("config")
    name = _name()
    digest = B.config_digest(model, data)
    seg = _ready_segment(name, digest)
    srv = StandInServer(seg)
    srv.start()
    made = []
    import victor_amd.engine as E
    monkeypatch.setattr(E.Engine, "__init__", lambda self, *a, **k: made.append(1) or (_ for _ in ()).throw(N.NativeError("no GPU")))
    try:
        fit = victor_amd.CCFFit(model, data, broker=name)
        p = {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0}
        lnl, chi2 = fit.log_likelihood(dict(p))
        row = fit._scalar_row(p, True, True)
        assert (lnl, chi2) == (sum(row), 2.0 * row[0]) and not made
        with pytest.raises(victor_amd.InputError):
            fit.log_likelihood({"fsigma8": 0.47, "sigma_v": 380, "epsilon": 1.0})         # beta missing: the reference's error
        with pytest.raises(N.NativeError):
            fit.log_likelihood(dict(p), rsd_model="kaiser")                                 # not the broker's plan: local context
        assert made
        fit._broker_client.close()
    finally:
        srv.stop = True
        srv.join(2)
        path = seg.path
        seg.close()
        for p_ in (path, path + ".lock"):
            if os.path.exists(p_):
                os.unlink(p_)
