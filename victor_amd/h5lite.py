"""Minimal pure-Python reader for the HDF5 files victor consumes.

The reference reads its model / data / covariance inputs with ``h5py``
(``ccf_model.py:64-68``, ``ccf_fit.py:53-57``, ``ccf_fit.py:125-129``): every
top-level dataset is pulled into a dict of NumPy arrays.  ``h5py`` is not
available in the target image, so this module walks the on-disk format
directly for exactly that use: a root group holding contiguous (or compact),
unfiltered, fixed-size numeric datasets, as written by h5py with its default
``libver='earliest'`` settings (superblock v0/v1, v1 object headers, symbol
table groups).  Anything else raises :class:`H5LiteError` with a message that
tells the user to install h5py, which :func:`read_all` prefers when present.
"""

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5LiteError(Exception):
    """The file uses an HDF5 feature outside the supported subset."""


class _Reader:
    def __init__(self, buf):
        self.buf = buf
        base = buf.find(_SIG)
        if base != 0:
            # the signature may sit at 512, 1024, ... for files with a user block
            pos = 512
            base = -1
            while pos < len(buf):
                if buf[pos:pos + 8] == _SIG:
                    base = pos
                    break
                pos *= 2
            if base < 0:
                raise H5LiteError("not an HDF5 file (signature missing)")
        self.sb = base
        ver = buf[base + 8]
        if ver not in (0, 1):
            raise H5LiteError(f"superblock version {ver} unsupported (install h5py)")
        self.osz = buf[base + 13]
        self.lsz = buf[base + 14]
        if self.osz != 8 or self.lsz != 8:
            raise H5LiteError("only 8-byte offsets/lengths supported")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr = self.u64(p)
        p += 4 * 8  # base, free-space, eof, driver-info
        self.root = self._symtab_entry(p)

    # --- primitive accessors -------------------------------------------------
    def u8(self, p):
        return self.buf[p]

    def u16(self, p):
        return struct.unpack_from("<H", self.buf, p)[0]

    def u32(self, p):
        return struct.unpack_from("<I", self.buf, p)[0]

    def u64(self, p):
        return struct.unpack_from("<Q", self.buf, p)[0]

    def addr(self, a):
        return self.base_addr + a

    # --- groups --------------------------------------------------------------
    def _symtab_entry(self, p):
        name_off = self.u64(p)
        ohdr = self.u64(p + 8)
        cache = self.u32(p + 16)
        btree = heap = None
        if cache == 1:
            btree = self.u64(p + 24)
            heap = self.u64(p + 32)
        return {"name_off": name_off, "ohdr": ohdr, "btree": btree, "heap": heap}

    def _heap_data(self, heap_addr):
        p = self.addr(heap_addr)
        if self.buf[p:p + 4] != b"HEAP":
            raise H5LiteError("bad local heap signature")
        return self.addr(self.u64(p + 8 + 16))

    def _cstr(self, p):
        end = self.buf.index(b"\x00", p)
        return self.buf[p:end].decode("utf-8")

    def _walk_btree(self, node_addr, heap_data, out):
        p = self.addr(node_addr)
        if self.buf[p:p + 4] != b"TREE":
            raise H5LiteError("bad B-tree node signature")
        ntype, level, used = self.u8(p + 4), self.u8(p + 5), self.u16(p + 6)
        if ntype != 0:
            raise H5LiteError("unexpected B-tree node type in group")
        q = p + 8 + 16  # skip sibling pointers
        for i in range(used):
            child = self.u64(q + 8 + i * 16)
            if level > 0:
                self._walk_btree(child, heap_data, out)
            else:
                self._read_snod(child, heap_data, out)

    def _read_snod(self, a, heap_data, out):
        p = self.addr(a)
        if self.buf[p:p + 4] != b"SNOD":
            raise H5LiteError("bad symbol-table node signature")
        n = self.u16(p + 6)
        q = p + 8
        for i in range(n):
            e = self._symtab_entry(q + 40 * i)
            out.append((self._cstr(heap_data + e["name_off"]), e["ohdr"]))

    def members(self):
        """List ``(name, object_header_address)`` for the root group."""
        btree, heap = self.root["btree"], self.root["heap"]
        if btree is None:
            for mtype, mp, _ in self._messages(self.root["ohdr"]):
                if mtype == 0x11:
                    btree, heap = self.u64(mp), self.u64(mp + 8)
        if btree is None:
            raise H5LiteError("root group has no symbol table (new-style group; install h5py)")
        out = []
        self._walk_btree(btree, self._heap_data(heap), out)
        return out

    # --- object headers ------------------------------------------------------
    def _messages(self, ohdr_addr):
        p = self.addr(ohdr_addr)
        if self.buf[p:p + 4] == b"OHDR":
            raise H5LiteError("version-2 object headers unsupported (install h5py)")
        if self.u8(p) != 1:
            raise H5LiteError("unsupported object header version")
        nmsg = self.u16(p + 2)
        hsize = self.u32(p + 8)
        blocks = [(p + 16, hsize)]
        seen = 0
        while blocks and seen < nmsg:
            q, size = blocks.pop(0)
            end = q + size
            while q + 8 <= end and seen < nmsg:
                mtype, msize = self.u16(q), self.u16(q + 2)
                data = q + 8
                seen += 1
                if mtype == 0x10:
                    blocks.append((self.addr(self.u64(data)), self.u64(data + 8)))
                else:
                    yield mtype, data, msize
                q = data + msize

    def dataset(self, ohdr_addr):
        """Decode the dataset at an object header; ``None`` if it is not one."""
        shape = dtype = layout = None
        for mtype, p, size in self._messages(ohdr_addr):
            if mtype == 0x01:
                ver, rank, flags = self.u8(p), self.u8(p + 1), self.u8(p + 2)
                q = p + (8 if ver == 1 else 4)
                shape = tuple(self.u64(q + 8 * i) for i in range(rank))
            elif mtype == 0x03:
                cls = self.u8(p) & 0x0F
                bits0 = self.u8(p + 1)
                nbytes = self.u32(p + 4)
                order = ">" if (bits0 & 1) else "<"
                if cls == 1:
                    dtype = np.dtype(f"{order}f{nbytes}")
                elif cls == 0:
                    signed = (bits0 >> 3) & 1
                    dtype = np.dtype(f"{order}{'i' if signed else 'u'}{nbytes}")
                else:
                    raise H5LiteError(f"datatype class {cls} unsupported (install h5py)")
            elif mtype == 0x08:
                ver = self.u8(p)
                if ver != 3:
                    raise H5LiteError(f"data layout version {ver} unsupported (install h5py)")
                lclass = self.u8(p + 1)
                if lclass == 1:
                    layout = ("contiguous", self.u64(p + 2), self.u64(p + 10))
                elif lclass == 0:
                    layout = ("compact", p + 4, self.u16(p + 2))
                else:
                    raise H5LiteError("chunked datasets unsupported (install h5py)")
            elif mtype == 0x0B:
                raise H5LiteError("filtered datasets unsupported (install h5py)")
        if shape is None or dtype is None or layout is None:
            return None
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        kind, a, nbytes = layout
        if kind == "contiguous":
            if a == _UNDEF:
                return np.zeros(shape, dtype=dtype.newbyteorder("="))
            start = self.addr(a)
        else:
            start = a
        arr = np.frombuffer(self.buf, dtype=dtype, count=count, offset=start)
        return arr.reshape(shape).astype(dtype.newbyteorder("="), copy=True)


def read_all(path):
    """Return ``{name: ndarray}`` for every dataset in the root group of ``path``.

    Mirrors the reference's ``for key in f.keys(): input_data[key] = f[key][:]``
    loop (``ccf_model.py:65-68``).  Uses h5py when it is importable.
    """
    try:
        import h5py  # noqa: F401
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(path, "r") as f:
            return {key: f[key][:] for key in f.keys()}
    with open(path, "rb") as fh:
        rd = _Reader(fh.read())
    out = {}
    for name, ohdr in rd.members():
        arr = rd.dataset(ohdr)
        if arr is not None:
            out[name] = arr
    return out
