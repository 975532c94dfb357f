"""CPU tests of the device-wide ledger of reserved waiters (victor_amd/csrc/vk_ledger.cpp; include/victor_hip.h: vk_ledger_*).

The ledger needs no GPU: it is a file, a handful of atomics and /proc.  The library exports its operations on a path of the
caller's choosing for an owner of the caller's making (development entry points: VICTOR_HIP_DEV=1), so every case the bound
depends on is driven here: a dead owner's slot is reclaimed, a recycled pid is not honoured, another pid namespace is judged
conservatively, the 127th claimant gets no slot, two copies of the library in one process keep a slot each, a file that is not
ours to trust means no polling at all."""

import ctypes as C
import os
import stat
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OPENED, UNAVAILABLE, UNTRUSTED, FULL = 0, 1, 2, 3
FOREIGN_NS = 0x7fff_0000_0001


@pytest.fixture(scope="module")
def lib():
    os.environ["VICTOR_HIP_DEV"] = "1"
    from victor_amd import _native
    from victor_amd.build import build_native
    build_native()
    return _native.load()


@pytest.fixture()
def path(tmp_path):
    return str(tmp_path / "ledger").encode()


class Owner:
    """One claimant: a handle of the library plus the process-wide count the library keeps beside it."""

    def __init__(self, lib, path, pid=0, start=0, ns=0, lib_id=0):
        self.lib = lib
        st = C.c_int32(-1)
        self.h = lib.vk_ledger_open_at(path, pid, start, ns, lib_id, C.byref(st))
        self.status = st.value
        self.process = C.c_int32(0)

    def grant(self, want, ctx_reserved=0):
        return self.lib.vk_ledger_grant(self.h, C.byref(self.process), ctx_reserved, want)

    def release(self, n):
        self.lib.vk_ledger_release(self.h, C.byref(self.process), n)

    def others(self):
        return self.lib.vk_ledger_others(self.h)

    def slot(self):
        return self.lib.vk_ledger_slot(self.h)

    def close(self, keep_slot=False):
        if self.h:
            self.lib.vk_ledger_close(self.h, 1 if keep_slot else 0)
        self.h = None


def me(lib):
    return tuple(int(lib.vk_ledger_self(k, 0)) for k in range(4))          # pid, start, ns, lib


def layout(lib):
    v = [C.c_int32() for _ in range(4)]
    lib.vk_ledger_layout(*[C.byref(x) for x in v])
    return tuple(x.value for x in v)                                       # header bytes, slot bytes, slots, version


def read_slots(lib, path):
    hdr, sb, n, _ = layout(lib)
    raw = open(path, "rb").read()
    assert len(raw) == hdr + sb * n
    return [struct.unpack_from("<qQQQii", raw, hdr + i * sb) for i in range(n)]      # pid, start, ns, lib, reserved, pad


def write_slot(lib, path, i, pid, start, ns, lib_id, reserved):
    hdr, sb, _, _ = layout(lib)
    with open(path, "r+b") as fh:
        fh.seek(hdr + i * sb)
        fh.write(struct.pack("<qQQQii", pid, start, ns, lib_id, reserved, 0))


def dead_pid():
    """pid of a process that has existed and is gone (reaped)."""
    p = subprocess.Popen([sys.executable, "-c", "pass"])
    p.wait()
    return p.pid


def test_the_entry_points_need_the_development_switch(lib, path):
    os.environ.pop("VICTOR_HIP_DEV")
    try:
        st = C.c_int32(-1)
        assert lib.vk_ledger_open_at(path, 0, 0, 0, 0, C.byref(st)) is None and not os.path.exists(path)
        assert lib.vk_ledger_others(None) == -1 and lib.vk_ledger_slot(None) == -1
    finally:
        os.environ["VICTOR_HIP_DEV"] = "1"


def test_layout_identity_and_a_first_claim(lib, path):
    assert layout(lib) == (16, 40, 126, 2)
    pid, start, ns, lib_id = me(lib)
    assert pid == os.getpid() and start > 0 and ns > 0 and lib_id > 0
    # field 22 of /proc/<pid>/stat is what the library calls the start time; the command name may contain ') '
    with open(f"/proc/{pid}/stat") as fh:
        assert start == int(fh.read().rsplit(")", 1)[1].split()[19])
    assert lib.vk_ledger_self(5, pid) == 1 and lib.vk_ledger_self(4, pid) == start
    gone = dead_pid()
    assert lib.vk_ledger_self(5, gone) == 0 and lib.vk_ledger_self(4, gone) == 0
    a = Owner(lib, path)
    assert a.status == OPENED and a.slot() == 0 and a.others() == 0
    mode = os.stat(path)
    assert stat.S_IMODE(mode.st_mode) == 0o600 and mode.st_size == 16 + 40 * 126
    magic, version, gen, _ = struct.unpack_from("<IIII", open(path, "rb").read())
    assert magic == 0x564b504c and version == 2 and gen >= 1
    assert read_slots(lib, path)[0][:4] == (pid, start, ns, lib_id)
    assert a.grant(5) == 5 and read_slots(lib, path)[0][4] == 5 and a.process.value == 5
    a.release(5)
    assert read_slots(lib, path)[0][4] == 0 and a.process.value == 0
    a.close()
    assert read_slots(lib, path)[0][0] == 0                          # a clean exit frees the slot


def test_a_dead_owners_slot_is_ignored_and_reclaimed(lib, path):
    pid, start, ns, lib_id = me(lib)
    a = Owner(lib, path)
    gone = dead_pid()
    write_slot(lib, path.decode(), 1, gone, 12345, ns, 77, 30)       # a process that died holding 30 waiters
    assert a.others() == 0 and a.grant(8) == 8                       # ... does not count
    b = Owner(lib, path, lib_id=lib_id + 1)                          # (another copy of the library in this process)
    assert b.status == OPENED and b.slot() == 2                      # free slots go first ...
    slots = read_slots(lib, path.decode())
    assert slots[1][0] == gone and slots[2][:4] == (pid, start, ns, lib_id + 1)
    # ... and when none is left the dead owner's slot changes hands, its reservation wiped
    for i in range(3, 126):
        write_slot(lib, path.decode(), i, pid, start, ns, 1000 + i, 0)
    c = Owner(lib, path, lib_id=lib_id + 2)
    assert c.status == OPENED and c.slot() == 1
    assert read_slots(lib, path.decode())[1] == (pid, start, ns, lib_id + 2, 0, 0)
    for o in (a, b, c):
        o.close()


def test_a_recycled_pid_does_not_inherit_a_reservation(lib, path):
    """A slot whose pid names a LIVING process with another start time: the writer is gone, the pid has been handed on."""
    pid, start, ns, lib_id = me(lib)
    sleeper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(60)"])
    try:
        real_start = int(lib.vk_ledger_self(4, sleeper.pid))
        assert real_start > 0
        a = Owner(lib, path)
        write_slot(lib, path.decode(), 1, sleeper.pid, real_start + 1, ns, 5, 40)      # same pid, an earlier life
        assert a.others() == 0
        write_slot(lib, path.decode(), 2, sleeper.pid, real_start, ns, 5, 40)          # the living process itself
        assert a.others() == 40 and a.grant(8) == 8 and a.grant(8, ctx_reserved=0) == 8 and a.grant(8) == 0      # 40 + 16 + 8 > 63
        # and the living one's slot is not for the taking, the stale one is
        for i in range(3, 126):
            write_slot(lib, path.decode(), i, pid, start, ns, 1000 + i, 0)
        b = Owner(lib, path, lib_id=lib_id + 1)
        assert b.status == OPENED and b.slot() == 1
        c = Owner(lib, path, lib_id=lib_id + 2)
        assert c.status == FULL and c.h is None
        a.close()
        b.close()
    finally:
        sleeper.kill()
        sleeper.wait()


def test_a_small_pid_of_another_namespace_is_not_adopted(lib, path):
    """Containers sharing /dev/shm: pid 7 of one is not pid 7 of the other.  A slot is one's own only if pid, start time,
    namespace and library instance all agree; a foreign slot counts as living whatever kill() says about its number here."""
    pid, start, ns, lib_id = me(lib)
    write_header_via = Owner(lib, path)                              # creates and stamps the file
    write_header_via.close()
    write_slot(lib, path.decode(), 0, pid, start, FOREIGN_NS, lib_id, 20)         # "me" by number, in another namespace
    write_slot(lib, path.decode(), 1, dead_pid(), 1, FOREIGN_NS, 1, 30)           # dead HERE - but judged from here it cannot be
    a = Owner(lib, path)
    assert a.status == OPENED and a.slot() == 2                      # neither adopted nor taken over
    assert a.others() == 50
    assert a.grant(8) == 8 and a.grant(8) == 0                       # 50 + 8 = 58: five left, all or nothing
    assert a.grant(5) == 5 and a.grant(1) == 0                       # 63 waiters on the device: the 64th never
    # foreign slots are never reclaimed, however full the ledger is
    for i in range(3, 126):
        write_slot(lib, path.decode(), i, pid, start, ns, 1000 + i, 0)
    b = Owner(lib, path, lib_id=lib_id + 1)
    assert b.status == FULL
    assert read_slots(lib, path.decode())[1][2] == FOREIGN_NS
    a.close()


def test_the_127th_claimant_gets_no_slot(lib, path):
    pid, start, ns, lib_id = me(lib)
    owners = [Owner(lib, path, lib_id=lib_id + k) for k in range(130)]
    assert [o.status for o in owners[:126]] == [OPENED] * 126 and sorted(o.slot() for o in owners[:126]) == list(range(126))
    assert [o.status for o in owners[126:]] == [FULL] * 4 and all(o.h is None for o in owners[126:])
    # 126 owners x 1 waiter would be 126: the device-wide bound holds among them
    granted = sum(o.grant(1) for o in owners[:126])
    assert granted == 63 and owners[0].others() == 62 and owners[125].others() == 63
    # an owner without a slot gets nothing (the library does not poll for it): grant on a NULL ledger is the no-ledger rule and
    # is never called for FULL / UNTRUSTED by the launch path (victor_hip.hip: may_ask)
    for o in owners[:126]:
        o.close()
    assert all(s[0] == 0 for s in read_slots(lib, path.decode()))


def test_two_copies_of_the_library_in_one_process_keep_a_slot_each(lib, path):
    """The product library and its development twin in one process (tests/devlib.py): each keeps its own count of reserved
    waiters; neither overwrites the other's published reservation, and each sees the other's in the device-wide sum."""
    pid, start, ns, lib_id = me(lib)
    prod, dev = Owner(lib, path), Owner(lib, path, lib_id=lib_id + 4096)
    assert prod.slot() != dev.slot()
    assert prod.grant(8) == 8 and prod.grant(8) == 8 and prod.grant(8) == 8 and prod.grant(8) == 8 and prod.grant(1) == 0      # its budget of 32
    assert dev.others() == 32 and prod.others() == 0
    again = Owner(lib, path)                                         # the same owner opening once more adopts its slot AS IT IS
    assert again.slot() == prod.slot() and read_slots(lib, path.decode())[prod.slot()][4] == 32
    assert dev.grant(8) == 8 and dev.grant(8) == 8 and dev.grant(8) == 8 and dev.grant(8) == 0       # 32 + 24 = 56; 64 would break the bound
    assert dev.grant(7) == 7 and dev.grant(1) == 0 and prod.others() == 31
    dev.release(31)
    prod.release(32)
    assert prod.others() == 0 and dev.others() == 0
    again.close(keep_slot=True)
    prod.close()
    dev.close()


def test_a_refused_owner_learns_from_the_generation_when_to_ask_again(lib, path):
    pid, start, ns, lib_id = me(lib)
    a, b = Owner(lib, path), Owner(lib, path, lib_id=lib_id + 1)
    for _ in range(4):
        assert a.grant(8) == 8
    for _ in range(3):
        assert b.grant(8) == 8
    assert b.grant(8) == 0
    g0 = lib.vk_ledger_generation(b.h)
    assert b.grant(8) == 0 and lib.vk_ledger_generation(b.h) == g0       # nothing has moved: asking again is pointless
    a.release(8)
    assert lib.vk_ledger_generation(b.h) != g0 and b.grant(8) == 8        # ... now it is worth asking: 24 + 24 + 8 = 56
    a.close()
    b.close()


def test_a_file_that_is_not_ours_to_trust_means_no_polling(lib, tmp_path):
    real = tmp_path / "real"
    # a symbolic link (somebody pre-created the predictable name pointing elsewhere)
    link = tmp_path / "link"
    a = Owner(lib, str(real).encode())
    a.close()
    os.symlink(real, link)
    o = Owner(lib, str(link).encode())
    assert o.status == UNTRUSTED and o.h is None
    # open to group or others: anybody may have edited the counts
    wide = tmp_path / "wide"
    Owner(lib, str(wide).encode()).close()
    os.chmod(wide, 0o666)
    assert Owner(lib, str(wide).encode()).status == UNTRUSTED
    os.chmod(wide, 0o600)
    assert Owner(lib, str(wide).encode()).status == OPENED
    # a file of another size, of another magic, of another version
    short = tmp_path / "short"
    short.write_bytes(b"\0" * 1016)                                  # (round 5's version-1 ledger was 1016 bytes)
    os.chmod(short, 0o600)
    assert Owner(lib, str(short).encode()).status == UNTRUSTED
    for name, head in (("magic", struct.pack("<II", 0x12345678, 2)), ("version", struct.pack("<II", 0x564b504c, 1))):
        f = tmp_path / name
        f.write_bytes(head + b"\0" * (16 + 40 * 126 - 8))
        os.chmod(f, 0o600)
        assert Owner(lib, str(f).encode()).status == UNTRUSTED, name
    # two hard links to the file: somebody else holds a name for it
    twin = tmp_path / "twin"
    os.link(real, twin)
    assert Owner(lib, str(real).encode()).status == UNTRUSTED
    os.unlink(twin)
    assert Owner(lib, str(real).encode()).status == OPENED
    # a directory in the way / no directory to create the file in: unavailable (the process budget alone applies)
    assert Owner(lib, str(tmp_path / "no" / "such" / "dir" / "ledger").encode()).status == UNAVAILABLE
    # a lock nobody releases
    import fcntl
    locked = tmp_path / "locked"
    Owner(lib, str(locked).encode()).close()
    with open(locked, "r+b") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        assert Owner(lib, str(locked).encode()).status == UNTRUSTED
    assert Owner(lib, str(locked).encode()).status == OPENED


_CLAIMANT = r'''
import ctypes as C, os, sys, time
sys.path.insert(0, sys.argv[1])
os.environ["VICTOR_HIP_DEV"] = "1"
from victor_amd import _native
lib = _native.load()
st = C.c_int32(-1)
h = lib.vk_ledger_open_at(sys.argv[2].encode(), 0, 0, 0, 0, C.byref(st))
process = C.c_int32(0)
got = sum(lib.vk_ledger_grant(h, C.byref(process), 8 * k, 8 * (k + 1)) for k in range(4)) if h else 0
print(st.value, lib.vk_ledger_slot(h) if h else -1, got, flush=True)
sys.stdin.readline()            # hold the reservations until told to go (or killed)
'''


def test_processes_racing_for_the_bound_never_pass_it(lib, path):
    """Four real processes each asking for their full budget at once: the sum the ledger ends up with is below 64 whatever the
    interleaving; a killed process's reservations stop counting the moment it is gone."""
    procs = [subprocess.Popen([sys.executable, "-c", _CLAIMANT, ROOT, path.decode()], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              text=True) for _ in range(4)]
    try:
        lines = [p.stdout.readline().split() for p in procs]
        assert all(ln[0] == str(OPENED) for ln in lines), lines
        assert sorted(int(ln[1]) for ln in lines) == [0, 1, 2, 3]
        got = [int(ln[2]) for ln in lines]
        assert all(g % 8 == 0 and g <= 32 for g in got) and 32 <= sum(got) <= 63, got
        a = Owner(lib, path)
        assert a.others() == sum(got)
        procs[0].kill()
        procs[0].wait()
        assert a.others() == sum(got) - got[0]                       # dead: its slot is ignored at once
        a.close()
    finally:
        for p in procs:
            if p.poll() is None:
                p.stdin.write("\n")
                p.stdin.flush()
                p.wait(timeout=20)


def test_ledger_operations_are_clean_under_the_sanitizers(tmp_path):
    """vk_ledger.cpp is plain POSIX C++ (no HIP): compiled on its own with -fsanitize=address,undefined and driven through the
    same operations (tests/helpers/asan_ledger.py) in an interpreter with the sanitizer runtimes preloaded - any report fails
    the run.  (GPU sanitizers are not available on the pool; this is the CPU build the sanitizers can see.)"""
    import shutil
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    rt = [subprocess.run([gxx, f"-print-file-name={n}"], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(r) and os.path.isfile(r) for r in rt):
        pytest.skip("sanitizer runtimes not installed")
    so = str(tmp_path / "libvk_ledger_san.so")
    build = subprocess.run([gxx, "-g", "-O1", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                            "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "victor_amd", "csrc", "vk_ledger.cpp"), "-o", so], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=":".join(rt), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", VICTOR_HIP_DEV="1")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "asan_ledger.py"), so], env=env, capture_output=True,
                         text=True, timeout=120)
    assert run.returncode == 0 and "clean" in run.stdout, (run.stdout[-1000:], run.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
