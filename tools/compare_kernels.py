"""Kernel by kernel, the instruction streams of two builds of the library (gfx950 code objects inside their fat binaries):
which kernels both hold, which differ, and in how many instructions.  Needs no GPU.
Usage: python tools/compare_kernels.py libA.so libB.so

Branch targets and the offsets of scalar literal loads differ with a kernel's position in its code object, so instructions are
compared with their addresses and pc-relative operands masked; everything else - opcode, registers, immediates, order - must
agree for a kernel to count as identical."""
import hashlib
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, tmp, tag):
    sec = subprocess.run([f"{LLVM}/llvm-readelf", "-S", lib], capture_output=True, text=True).stdout
    m = re.search(r"\.hip_fatbin\s+PROGBITS\s+[0-9a-f]+\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
    with open(lib, "rb") as fh:
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
                path = os.path.join(tmp, f"{tag}_{len(paths)}.co")
                with open(path, "wb") as fh:
                    fh.write(blob[at + off:at + off + size])
                paths.append(path)
        start = at + len(magic)
    return paths


def normalise(ins):
    ins = re.sub(r"<[^>]*>", "<sym>", ins)                              # branch targets by symbol + offset
    ins = re.sub(r"\b(s_c?branch\w*|s_call\w*)\s+\d+", r"\1 <rel>", ins)  # ... and by number
    return ins


def kernels(lib, tmp, tag):
    """{kernel: (instruction count, digest of the normalised stream)}; kernels only (what the notes list), not helper symbols."""
    out = {}
    for co in code_objects(lib, tmp, tag):
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        names = set(re.findall(r"\.name:\s+(\S+)", notes))
        asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
        cur, body = None, []

        def close():
            if cur in names:
                text = [normalise(b) for b in body if not b.startswith("s_nop") and not b.startswith("s_code_end")]
                out[cur] = (len(text), hashlib.sha256("\n".join(text).encode()).hexdigest(), text)
        for ln in asm.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                if re.match(r"^L?BB\d+_\d+$", m.group(1)) or m.group(1).startswith(".L"):
                    continue                                             # a label inside the current kernel
                close()
                cur, body = m.group(1), []
            elif ln.strip() and cur is not None:
                body.append(ln.split("//")[0].strip())
        close()
    return out


def main():
    a_lib, b_lib = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as tmp:
        a, b = kernels(a_lib, tmp, "a"), kernels(b_lib, tmp, "b")
    both = sorted(set(a) & set(b))
    same = [k for k in both if a[k][1] == b[k][1]]
    differ = [k for k in both if a[k][1] != b[k][1]]
    print(f"A: {a_lib}: {len(a)} kernels, {sum(v[0] for v in a.values())} instructions")
    print(f"B: {b_lib}: {len(b)} kernels, {sum(v[0] for v in b.values())} instructions")
    print(f"in both: {len(both)}; identical instruction streams: {len(same)}; different: {len(differ)}")
    print(f"only in A: {len(set(a) - set(b))}; only in B: {len(set(b) - set(a))}")
    for k in sorted(set(a) - set(b))[:20]:
        print("  only A:", subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip())
    for k in sorted(set(b) - set(a))[:20]:
        print("  only B:", subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip())
    for k in differ[:40]:
        ta, tb = a[k][2], b[k][2]
        n_diff = sum(1 for x, y in zip(ta, tb) if x != y) + abs(len(ta) - len(tb))
        first = next((i for i, (x, y) in enumerate(zip(ta, tb)) if x != y), min(len(ta), len(tb)))
        print(f"  differs: {subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()}: {len(ta)} vs {len(tb)} instructions, "
              f"{n_diff} positions differ, first at {first}: {ta[first] if first < len(ta) else '-'} | {tb[first] if first < len(tb) else '-'}")
    return 0 if not differ else 1


if __name__ == "__main__":
    sys.exit(main())
