"""Minimal pure-Python reader for the HDF5 flavour the reference's case files use (host only, no h5py in this image).

The reference stores its cases with HDF5.jl defaults (src/powerSystem/save.jl; read back by src/powerSystem/load.jl:
36-67, 141-289, 1360-1367): superblock version 0, version-1 object headers, old-style groups (symbol-table message ->
version-1 B-tree of symbol nodes + local heap), CONTIGUOUS (or compact) datasets without filters, little-endian
fixed-point / IEEE floating-point element types.  That subset of the published HDF5 file format specification
(version 3.0, sections II.A superblock, III.A B-trees, III.C symbol nodes, III.D local heaps, IV.A object headers and the
dataspace / datatype / layout / continuation / symbol-table messages) is what is parsed here; anything else raises
NotImplementedError rather than guessing.  String datasets (labels) are reported as None.
"""
from __future__ import annotations

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File:
    def __init__(self, path: str):
        with open(path, "rb") as f:
            self.buf = f.read()
        b = self.buf
        if b[:8] != _SIG:
            raise ValueError(f"{path}: not an HDF5 file")
        if b[8] != 0:
            raise NotImplementedError(f"{path}: superblock version {b[8]} (only 0 is supported)")
        if b[13] != 8 or b[14] != 8:
            raise NotImplementedError("only 8-byte offsets and lengths are supported")
        self.base = struct.unpack_from("<Q", b, 24)[0]
        # root group symbol table entry at 24 + 4 * 8: link name offset, object header address, cache type, -, scratch pad
        _, hdr, cache = struct.unpack_from("<QQI", b, 56)
        self.root = hdr
        self._datasets: dict[str, int] = {}
        self.skipped: list[str] = []                    # groups this reader cannot enumerate (dense link storage)
        self._walk("", hdr)

    # -- object headers ---------------------------------------------------------------------------------------------
    def _messages(self, addr: int):
        """(type, data offset, size) of every header message of a version-1 object header, continuations included."""
        b = self.buf
        a = self.base + addr
        if b[a] != 1:
            raise NotImplementedError(f"object header version {b[a]} (only 1 is supported)")
        nmsg, = struct.unpack_from("<H", b, a + 2)
        size, = struct.unpack_from("<I", b, a + 8)
        blocks = [(a + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                data = pos + 8
                if mtype == 0x0010:                                     # object header continuation: offset, length
                    off, ln = struct.unpack_from("<QQ", b, data)
                    blocks.append((self.base + off, ln))
                out.append((mtype, data, msize))
                pos = data + msize
        return out

    def _walk(self, prefix: str, addr: int):
        b = self.buf
        msgs = self._messages(addr)
        sym = [m for m in msgs if m[0] == 0x0011]
        links = [m for m in msgs if m[0] == 0x0006]
        if not sym and (links or any(m[0] == 0x0002 for m in msgs)):    # new-style group with COMPACT link storage
            for m in msgs:
                if m[0] == 0x0002:                                      # link info: a fractal heap address means dense storage
                    flags = b[m[1] + 1]
                    q = m[1] + 2 + (8 if flags & 1 else 0)
                    heap, = struct.unpack_from("<Q", b, q)
                    if heap != _UNDEF:                                  # dense storage: the links live in a fractal heap
                        try:
                            for name, hdr in self._dense_links(heap):
                                self._walk(prefix + "/" + name, hdr)
                        except NotImplementedError:
                            self.skipped.append(prefix or "/")
                        return
            for _, q, _size in links:
                if b[q] != 1:
                    raise NotImplementedError("link message version")
                flags = b[q + 1]
                q += 2
                ltype = 0
                if flags & 8:
                    ltype = b[q]; q += 1
                if flags & 4:
                    q += 8
                if flags & 16:
                    q += 1
                w = 1 << (flags & 3)
                ln = int.from_bytes(b[q:q + w], "little"); q += w
                name = b[q:q + ln].decode(); q += ln
                if ltype != 0:
                    continue                                            # soft / external links: not used by the case files
                hdr, = struct.unpack_from("<Q", b, q)
                self._walk(prefix + "/" + name, hdr)
            return
        if not sym:                                                     # a dataset (or a named datatype): remember it
            self._datasets[prefix or "/"] = addr
            return
        btree, heap = struct.unpack_from("<QQ", b, sym[0][1])
        heap_a = self.base + heap
        if b[heap_a:heap_a + 4] != b"HEAP":
            raise ValueError("local heap signature missing")
        seg, = struct.unpack_from("<Q", b, heap_a + 24)
        seg += self.base

        def name_at(off):
            e = b.index(b"\x00", seg + off)
            return b[seg + off:e].decode()

        def node(a):
            a += self.base
            if b[a:a + 4] == b"TREE":
                ntype, level, used = struct.unpack_from("<BBH", b, a + 4)
                if ntype != 0:
                    raise NotImplementedError("only group B-trees are walked")
                p = a + 8 + 16                                          # after the two sibling pointers
                for k in range(used):
                    child, = struct.unpack_from("<Q", b, p + 8 + k * 16)   # key0, child0, key1, child1, ...
                    node(child)
            elif b[a:a + 4] == b"SNOD":
                nsym, = struct.unpack_from("<H", b, a + 6)
                for k in range(nsym):
                    off, hdr = struct.unpack_from("<QQ", b, a + 8 + k * 40)
                    self._walk(prefix + "/" + name_at(off), hdr)
            else:
                raise ValueError("unknown group node signature")

        node(btree)

    def _link(self, q: int):
        """One link message at q -> (name, object header address or None, end offset)."""
        b = self.buf
        if b[q] != 1:
            raise NotImplementedError("link message version")
        flags = b[q + 1]
        q += 2
        ltype = 0
        if flags & 8:
            ltype = b[q]; q += 1
        if flags & 4:
            q += 8
        if flags & 16:
            q += 1
        w = 1 << (flags & 3)
        ln = int.from_bytes(b[q:q + w], "little"); q += w
        name = b[q:q + ln].decode(); q += ln
        if ltype != 0:
            raise NotImplementedError("soft / external links")
        hdr, = struct.unpack_from("<Q", b, q)
        return name, hdr, q + 8

    def _dense_links(self, heap: int):
        """Links of a group with dense storage: the managed objects of its fractal heap (format spec III.G), read block by
        block in allocation order -- the writer appended them and never deleted one, so every direct block is filled from
        its start.  Filtered heaps, huge and tiny objects are not used by link heaps of this size."""
        b = self.buf
        a = self.base + heap
        if b[a:a + 4] != b"FRHP" or b[a + 4] != 0:
            raise NotImplementedError("fractal heap header")
        filt_len, flags = struct.unpack_from("<HB", b, a + 7)
        if filt_len:
            raise NotImplementedError("filtered fractal heap")
        nobj, = struct.unpack_from("<Q", b, a + 70)
        width, start, maxdirect, maxbits, _start_rows, root, cur_rows = struct.unpack_from("<HQQHHQH", b, a + 110)
        off_bytes = (maxbits + 7) // 8
        head = 4 + 1 + 8 + off_bytes + (4 if flags & 2 else 0)

        def direct_blocks():
            if cur_rows == 0:
                yield self.base + root, start
                return
            r = self.base + root
            if b[r:r + 4] != b"FHIB":
                raise NotImplementedError("fractal heap indirect block")
            p = r + 4 + 1 + 8 + off_bytes
            for row in range(cur_rows):
                size = start if row < 2 else start << (row - 1)
                if size > maxdirect:
                    raise NotImplementedError("nested indirect blocks")
                for _ in range(width):
                    child, = struct.unpack_from("<Q", b, p)
                    p += 8
                    if child != _UNDEF:
                        yield self.base + child, size

        out = []
        for blk, size in direct_blocks():
            if b[blk:blk + 4] != b"FHDB":
                raise NotImplementedError("fractal heap direct block")
            q, end = blk + head, blk + size
            while len(out) < nobj and q < end and b[q] == 1:
                name, hdr, q = self._link(q)
                out.append((name, hdr))
        if len(out) != nobj:
            raise NotImplementedError("fractal heap layout")
        return out

    # -- datasets ---------------------------------------------------------------------------------------------------
    def datasets(self):
        return sorted(self._datasets)

    def __contains__(self, name: str) -> bool:
        return name in self._datasets

    def read(self, name: str):
        """The dataset as a numpy array (0-d datasets as shape (1,)); None for element types other than plain numbers."""
        b = self.buf
        shape = dtype = None
        data = None
        for mtype, p, msize in self._messages(self._datasets[name]):
            if mtype == 0x0001:                                         # dataspace
                ver, rank, flags = b[p], b[p + 1], b[p + 2]
                q = p + (8 if ver == 1 else 4)
                shape = struct.unpack_from("<%dQ" % rank, b, q) if rank else ()
            elif mtype == 0x0003:                                       # datatype
                cls = b[p] & 0x0F
                bits0 = b[p + 1]
                size, = struct.unpack_from("<I", b, p + 4)
                if cls in (0, 1) and bits0 & 1:
                    raise NotImplementedError("big-endian data")
                if cls == 0:
                    dtype = np.dtype("<%s%d" % ("i" if bits0 & 8 else "u", size))
                elif cls == 1:
                    dtype = np.dtype("<f%d" % size)
                elif cls == 4:                                          # bitfield (Bool): one byte per element
                    dtype = np.dtype("u%d" % size)
                else:
                    dtype = None                                        # strings, compounds, ...: not needed for case tables
            elif mtype == 0x0008:                                       # data layout
                ver = b[p]
                if ver != 3:
                    raise NotImplementedError(f"data layout message version {ver}")
                lclass = b[p + 1]
                if lclass == 1:                                         # contiguous: address, size
                    addr, size = struct.unpack_from("<QQ", b, p + 2)
                    data = (None, 0) if addr == _UNDEF else (self.base + addr, size)
                elif lclass == 0:                                       # compact: size, data
                    size, = struct.unpack_from("<H", b, p + 2)
                    data = (p + 4, size)
                else:
                    raise NotImplementedError("chunked / virtual datasets are not supported")
            elif mtype == 0x000B:
                raise NotImplementedError("filtered datasets are not supported")
        if dtype is None:
            return None
        if data is None or shape is None:
            raise ValueError(f"{name}: incomplete dataset header")
        count = int(np.prod(shape)) if shape else 1
        if data[0] is None:
            return np.zeros(count, dtype=dtype)
        return np.frombuffer(b, dtype=dtype, count=count, offset=data[0]).copy()


def case_tables(path: str) -> dict:
    """Case tables of a reference HDF5 case file with the loader's conventions (load.jl:141-289, 1360-1367): a dataset
    with ONE element is a value shared by every element of its container ("compressed" scalar), from / to / generator
    bus are 1-based internal indices, everything is already per-unit / radians."""
    f = H5File(path)

    def g(name):
        a = f.read(name)
        if a is None:
            raise ValueError(f"{path}: dataset {name} is not numeric")
        return a

    def bc(name, n, dtype=None):
        a = g(name)
        a = np.full(n, a[0], dtype=a.dtype) if a.size == 1 and n != 1 else a
        if a.size != n:
            raise ValueError(f"{path}: {name} has {a.size} elements, expected {n} or 1")
        return np.ascontiguousarray(a.astype(dtype) if dtype is not None else a)

    n = g("/bus/layout/type").size
    frm = g("/branch/layout/from")
    nb = frm.size
    gbus = g("/generator/layout/bus")
    ng = gbus.size
    return dict(
        base_power=np.float64(g("/base/power")[0]),
        bus_type=bc("/bus/layout/type", n, np.int8),
        bus_pd=bc("/bus/demand/active", n), bus_qd=bc("/bus/demand/reactive", n),
        bus_gs=bc("/bus/shunt/conductance", n), bus_bs=bc("/bus/shunt/susceptance", n),
        bus_vm=bc("/bus/voltage/magnitude", n), bus_va=bc("/bus/voltage/angle", n),
        br_from=frm.astype(np.int64), br_to=g("/branch/layout/to").astype(np.int64),
        br_status=bc("/branch/layout/status", nb, np.int8),
        br_r=bc("/branch/parameter/resistance", nb), br_x=bc("/branch/parameter/reactance", nb),
        br_g=bc("/branch/parameter/conductance", nb), br_b=bc("/branch/parameter/susceptance", nb),
        br_tap=bc("/branch/parameter/turnsRatio", nb), br_shift=bc("/branch/parameter/shiftAngle", nb),
        gen_bus=gbus.astype(np.int64), gen_status=bc("/generator/layout/status", ng, np.int8),
        gen_pg=bc("/generator/output/active", ng), gen_qg=bc("/generator/output/reactive", ng),
        gen_vg=bc("/generator/voltage/magnitude", ng),
        gen_qmin=bc("/generator/capability/minReactive", ng), gen_qmax=bc("/generator/capability/maxReactive", ng),
    )
