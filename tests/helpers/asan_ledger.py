"""Driven by tests/test_ledger.py::test_ledger_operations_are_clean_under_the_sanitizers: the ledger's operations (claims up to a full
file, grants up to the bound, raw dead / foreign slots, releases, the untrusted and unavailable paths) on a build of vk_ledger.cpp
with -fsanitize=address,undefined, loaded into an interpreter that has the sanitizer runtimes preloaded.
Usage: asan_ledger.py <sanitized shared object>"""
import ctypes as C, os, struct, subprocess, sys
os.environ["VICTOR_HIP_DEV"] = "1"
lib = C.CDLL(sys.argv[1])
lib.vk_ledger_open_at.restype = C.c_void_p
lib.vk_ledger_open_at.argtypes = [C.c_char_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_int32)]
for n in ("vk_ledger_slot", "vk_ledger_others"):
    getattr(lib, n).argtypes = [C.c_void_p]; getattr(lib, n).restype = C.c_int32
lib.vk_ledger_grant.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32]; lib.vk_ledger_grant.restype = C.c_int32
lib.vk_ledger_release.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
lib.vk_ledger_close.argtypes = [C.c_void_p, C.c_int32]
lib.vk_ledger_self.argtypes = [C.c_int32, C.c_int64]; lib.vk_ledger_self.restype = C.c_uint64
import tempfile
d = tempfile.mkdtemp()
path = os.path.join(d, "ledger").encode()
me_lib = lib.vk_ledger_self(3, 0)
owners = []
for k in range(130):
    st = C.c_int32()
    h = lib.vk_ledger_open_at(path, 0, 0, 0, me_lib + k, C.byref(st))
    owners.append((h, st.value))
assert [s for _, s in owners[:126]] == [0] * 126 and [s for _, s in owners[126:]] == [3] * 4
tot = 0
for h, _ in owners[:126]:
    p = C.c_int32(0)
    tot += lib.vk_ledger_grant(h, C.byref(p), 0, 1)
assert tot == 63, tot
# a dead owner, a foreign namespace, a recycled pid, written raw
for h, _ in owners[:126]:
    lib.vk_ledger_close(h, 0)
p = subprocess.Popen([sys.executable, "-c", "pass"]); p.wait()
with open(path, "r+b") as fh:
    fh.seek(16); fh.write(struct.pack("<qQQQii", p.pid, 5, lib.vk_ledger_self(2, 0), 9, 30, 0))
    fh.seek(16 + 40); fh.write(struct.pack("<qQQQii", 7, 5, 12345, 9, 20, 0))
st = C.c_int32()
h = lib.vk_ledger_open_at(path, 0, 0, 0, 0, C.byref(st))
assert st.value == 0 and lib.vk_ledger_others(h) == 20 and lib.vk_ledger_slot(h) == 2
pr = C.c_int32(0)
assert lib.vk_ledger_grant(h, C.byref(pr), 0, 8) == 8
lib.vk_ledger_release(h, C.byref(pr), 8)
lib.vk_ledger_close(h, 0)
# untrusted / unavailable paths
os.symlink(path, path + b".lnk")
assert lib.vk_ledger_open_at(path + b".lnk", 0, 0, 0, 0, C.byref(st)) is None and st.value == 2
assert lib.vk_ledger_open_at(os.path.join(d, "no", "dir").encode(), 0, 0, 0, 0, C.byref(st)) is None and st.value == 1
print("asan/ubsan run of the ledger operations: clean")
