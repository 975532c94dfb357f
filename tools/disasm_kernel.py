"""Disassemble one kernel of libvictor_hip.so (gfx950 code objects inside the fat binary) and print instruction statistics.
Usage: python tools/disasm_kernel.py <substring of the mangled kernel name> [--dump]"""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("VICTOR_HIP_LIB", os.path.join(ROOT, "victor_amd", "csrc", "libvictor_hip.so"))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(tmp):
    sec = subprocess.run([f"{LLVM}/llvm-readelf", "-S", LIB], capture_output=True, text=True).stdout
    m = re.search(r"\.hip_fatbin\s+PROGBITS\s+[0-9a-f]+\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
    with open(LIB, "rb") as fh:
        fh.seek(int(m.group(1), 16))
        blob = fh.read(int(m.group(2), 16))
    magic, paths, start = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        at = blob.find(magic, start)
        if at < 0:
            break
        n, pos = struct.unpack_from("<Q", blob, at + 24)[0], at + 32
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, pos)
            ident = blob[pos + 24:pos + 24 + idlen].decode()
            pos += 24 + idlen
            if "gfx950" in ident:
                path = os.path.join(tmp, f"co_{len(paths)}.co")
                with open(path, "wb") as fh:
                    fh.write(blob[at + off:at + off + size])
                paths.append(path)
        start = at + len(magic)
    return paths


def main():
    want = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(tmp):
            asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
            cur, body = None, []
            for ln in asm.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    if cur and want in cur:
                        report(cur, body)
                    cur, body = m.group(1), []
                elif ln.strip():
                    body.append(ln.split("//")[0].strip())
            if cur and want in cur:
                report(cur, body)


def report(name, body):
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    ops = collections.Counter(ln.split()[0] for ln in body if ln and not ln.endswith(":"))
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    print(f"{demangled}: {len(body)} lines, v_* {valu}, ds_* {sum(v for k, v in ops.items() if k.startswith('ds_'))}, "
          f"s_* {sum(v for k, v in ops.items() if k.startswith('s_'))}")
    print("  ", ", ".join(f"{k} {v}" for k, v in ops.most_common(28)))
    if "--dump" in sys.argv:
        print("\n".join(body))


if __name__ == "__main__":
    main()
