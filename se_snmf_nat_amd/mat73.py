"""MATLAB `-v7.3` MAT-files (HDF5) for the basis dictionaries -- run_basis_train.m:136 (`save ... -v7.3`) and :138 (`load`).

A dictionary file holds four real double matrices (B_DFT_sub, B_Mel_sub, A_DFT_sub, A_Mel_sub; B_D_u.mat of
src/NTF_sep_event_RT.m:28-38,137-139 holds two).  h5py is not available where this runs, so this is a minimal reader / writer
of exactly that subset of the HDF5 file format (HDF5 File Format Specification, version 2.0):

  read    the 512-byte MATLAB user block; superblock version 0 (what MATLAB's HDF5 1.8 writes) or 2 / 3; the root group as a
          symbol table (v1 B-tree + local heap + symbol nodes) or as link messages in a version-2 object header; object headers
          version 1 and 2 incl. continuation blocks; dataspace v1 / v2; IEEE little-endian float64 / float32 and 1-, 2-, 4-,
          8-byte integers (MATLAB logical / char / intN classes); contiguous, compact and chunked layouts (v1 B-tree chunk
          index) with the deflate and shuffle filters -- `save -v7.3` compresses by default; the MATLAB_class attribute
  write   user block + superblock v0 + one symbol-table group + contiguous float64 datasets with MATLAB_class = "double"
          (the layout MATLAB itself uses for small uncompressed variables; MATLAB's `load` and h5dump read it)

A MATLAB m x n array is stored as an HDF5 dataset of dimensions (n, m): HDF5 is row-major, MATLAB column-major, the bytes are
the same.  Anything outside the subset (cell / struct variables, references, sparse, complex) raises ValueError naming it.
tests/test_mat73.py validates both directions against the HDF5 library's own tools where they exist (h5dump, h5repack).
"""
from __future__ import annotations

import struct
import time
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"


# ------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _ohdr_v1(msgs):
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


_F64_TYPE = struct.pack("<B3BI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)


def _dataset_header(dims, data_addr, nbytes):
    space = struct.pack("<BBB5x", 1, len(dims), 0) + b"".join(struct.pack("<Q", d) for d in dims)
    fill = struct.pack("<BBBB", 2, 2, 2, 0)  # version 2, late allocation, write if set, no value defined
    layout = struct.pack("<BBQQ", 3, 1, data_addr, nbytes)
    name = _pad8(b"MATLAB_class\0")
    atype = _pad8(struct.pack("<B3BI", 0x13, 0x00, 0x00, 0x00, 6))  # fixed-length string, null-terminated, ASCII, 6 bytes
    aspace = _pad8(struct.pack("<BBB5x", 1, 0, 0))                  # scalar
    attr = struct.pack("<BxHHH", 1, 13, 8, 8) + name + atype + aspace + b"double"
    return _ohdr_v1([_msg(0x0001, space), _msg(0x0003, _F64_TYPE, flags=1), _msg(0x0005, fill), _msg(0x0008, layout), _msg(0x000C, attr)])


def save_mat73(path, variables):
    """Write real double variables (scalars, vectors, 2-D arrays) as a MATLAB -v7.3 file."""
    names = sorted(variables)  # symbol nodes are ordered by name
    for nm in names:
        if not (nm.isidentifier() and len(nm) < 64):
            raise ValueError(f"'{nm}' is not a MATLAB variable name")
    arrs = {}
    for nm in names:
        a = np.asarray(variables[nm], dtype=np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(1, -1)  # MATLAB row vector
        if a.ndim != 2:
            raise ValueError(f"{nm}: only 2-D real double arrays are supported")
        arrs[nm] = a
    leaf_k = max(4, (len(names) + 1) // 2)       # a symbol node holds 2 * leaf_k entries: one node for everything
    internal_k = 16
    # local heap data segment: "" at offset 0, then the names
    heap = bytearray(b"\0" * 8)
    name_off = {}
    for nm in names:
        name_off[nm] = len(heap)
        heap += _pad8(nm.encode() + b"\0")
    heap_seg = bytes(heap)
    # layout (addresses relative to the superblock = base address)
    sb_size = 96
    root_hdr_addr = sb_size
    root_hdr = _ohdr_v1([_msg(0x0011, struct.pack("<QQ", 0, 0))])  # (addresses patched below)
    btree_addr = root_hdr_addr + len(root_hdr)
    btree_size = 24 + (2 * internal_k + 1) * 8 + 2 * internal_k * 8
    heap_addr = btree_addr + btree_size
    heap_hdr_size = 32
    heap_seg_addr = heap_addr + heap_hdr_size
    snod_addr = heap_seg_addr + len(heap_seg)
    snod_size = 8 + 2 * leaf_k * 40
    cursor = snod_addr + snod_size
    hdr_addr, data_addr, hdrs = {}, {}, {}
    for nm in names:
        a = arrs[nm]
        hdr_addr[nm] = cursor
        h = _dataset_header((a.shape[1], a.shape[0]), 0, a.size * 8)
        cursor += len(h)
        data_addr[nm] = cursor
        hdrs[nm] = _dataset_header((a.shape[1], a.shape[0]), cursor if a.size else UNDEF, a.size * 8)
        cursor += -(-a.size * 8 // 8) * 8
    eof = cursor
    out = bytearray()
    # superblock version 0
    out += SIG + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", leaf_k, internal_k, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)  # base address (see module docstring: relative addressing), free space, EOF, driver
    out += struct.pack("<QQII", 0, root_hdr_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr)  # root symbol table entry (cached)
    assert len(out) == sb_size
    out += _ohdr_v1([_msg(0x0011, struct.pack("<QQ", btree_addr, heap_addr))])
    # group B-tree: one leaf level, one child (the symbol node); key 0 = "", key 1 = the last name
    bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, name_off[names[-1]] if names else 0)
    out += bt + b"\0" * (btree_size - len(bt))
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_seg), 1, heap_seg_addr) + heap_seg  # free-list head 1 = no free block
    sn = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for nm in names:
        sn += struct.pack("<QQII16x", name_off[nm], hdr_addr[nm], 0, 0)
    out += sn + b"\0" * (snod_size - len(sn))
    for nm in names:
        assert len(out) == hdr_addr[nm]
        out += hdrs[nm]
        out += _pad8(np.asfortranarray(arrs[nm]).tobytes(order="F"))
    assert len(out) == eof
    # the MATLAB user block: 116 bytes of text, 8 bytes subsystem offset, version 0x0200, endian indicator "IM"; 512 bytes in all
    text = ("MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: %s HDF5 schema 1.00 ." % time.strftime("%a %b %d %H:%M:%S %Y")).encode()
    ub = text.ljust(116, b" ")[:116] + b"\0" * 8 + struct.pack("<H", 0x0200) + b"IM"
    with open(path, "wb") as f:
        f.write(ub.ljust(512, b"\0"))
        f.write(bytes(out))


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
class _File:
    def __init__(self, buf):
        self.buf = buf
        self.base = None
        for off in (0, 512, 1024, 2048, 4096):  # the superblock sits at 0 or at a power of two >= 512 (user block)
            if buf[off:off + 8] == SIG:
                self.base = off
                break
        if self.base is None:
            raise ValueError("not an HDF5 / MATLAB -v7.3 file (no superblock signature)")
        b = self.base
        ver = buf[b + 8]
        if ver in (0, 1):
            self.so, self.sl = buf[b + 13], buf[b + 14]
            p = b + 24 + (4 if ver == 1 else 0)
            if (self.so, self.sl) != (8, 8):
                raise ValueError("only 8-byte offsets / lengths are supported")
            p += 32  # base address, free space, EOF, driver info
            _name, self.root_hdr, cache, _r = struct.unpack_from("<QQII", buf, p)
        elif ver in (2, 3):
            self.so, self.sl = buf[b + 9], buf[b + 10]
            if (self.so, self.sl) != (8, 8):
                raise ValueError("only 8-byte offsets / lengths are supported")
            _base, _ext, _eof, self.root_hdr = struct.unpack_from("<QQQQ", buf, b + 12)
        else:
            raise ValueError(f"superblock version {ver} is not supported")

    def at(self, addr):
        return self.base + addr

    # ---- object headers -> list of (type, flags, bytes) ------------------------------------------
    def messages(self, addr):
        buf, p = self.buf, self.at(addr)
        out = []
        if buf[p:p + 4] == b"OHDR":  # version 2
            flags = buf[p + 5]
            q = p + 6
            if flags & 0x20:
                q += 16
            if flags & 0x10:
                q += 4
            szb = 1 << (flags & 3)
            chunk0 = int.from_bytes(buf[q:q + szb], "little")
            q += szb
            blocks = [(q, chunk0)]
            track = bool(flags & 0x04)
            while blocks:
                q, n = blocks.pop(0)
                end = q + n
                while q + 4 <= end:
                    mt, ms, mf = buf[q], struct.unpack_from("<H", buf, q + 1)[0], buf[q + 3]
                    q += 4 + (2 if track else 0)
                    data = bytes(buf[q:q + ms])
                    q += ms
                    if mt == 0x10:
                        ca, cl = struct.unpack_from("<QQ", data)
                        blocks.append((self.at(ca) + 4, cl - 8))  # skip "OCHK", drop the checksum
                    elif mt != 0:
                        out.append((mt, mf, data))
            return out
        ver, nmsg, _ref, hsize = struct.unpack_from("<BxHII", buf, p)
        if ver != 1:
            raise ValueError(f"object header version {ver} at {addr} is not supported")
        blocks = [(p + 16, hsize)]
        while blocks and len(out) < nmsg + 64:
            q, n = blocks.pop(0)
            end = q + n
            while q + 8 <= end:
                mt, ms, mf = struct.unpack_from("<HHB", buf, q)
                data = bytes(buf[q + 8:q + 8 + ms])
                q += 8 + ms
                if mt == 0x10:
                    ca, cl = struct.unpack_from("<QQ", data)
                    blocks.append((self.at(ca), cl))
                elif mt != 0:
                    out.append((mt, mf, data))
        return out

    # ---- groups ----------------------------------------------------------------------------------
    def group_members(self, addr):
        members = {}
        for mt, _mf, d in self.messages(addr):
            if mt == 0x11:  # symbol table message: v1 B-tree + local heap
                bt, hp = struct.unpack_from("<QQ", d)
                hb = self.at(hp)
                if self.buf[hb:hb + 4] != b"HEAP":
                    raise ValueError("bad local heap")
                seg = self.at(struct.unpack_from("<Q", self.buf, hb + 24)[0])
                self._walk_group_btree(bt, seg, members)
            elif mt == 0x06:  # link message (new-style compact group)
                ver, fl = d[0], d[1]
                q = 2
                ltype = 0
                if fl & 0x08:
                    ltype = d[q]
                    q += 1
                if fl & 0x04:
                    q += 8
                if fl & 0x10:
                    q += 1
                lsz = 1 << (fl & 3)
                ln = int.from_bytes(d[q:q + lsz], "little")
                q += lsz
                name = d[q:q + ln].decode()
                q += ln
                if ltype == 0:
                    members[name] = struct.unpack_from("<Q", d, q)[0]
            elif mt == 0x02:  # link info: compact groups carry their links as messages (above); dense ones in a fractal heap
                q = 2 + (8 if d[1] & 1 else 0)
                if struct.unpack_from("<Q", d, q)[0] != UNDEF:
                    raise ValueError("dense link storage (fractal heap) is not supported: the group has too many members")
        return members

    def _walk_group_btree(self, addr, heap_seg, members):
        p = self.at(addr)
        buf = self.buf
        if buf[p:p + 4] != b"TREE":
            raise ValueError("bad group B-tree node")
        ntype, level, used = struct.unpack_from("<BBH", buf, p + 4)
        q = p + 24
        for i in range(used):
            child = struct.unpack_from("<Q", buf, q + 8 + 16 * i)[0]
            if level > 0:
                self._walk_group_btree(child, heap_seg, members)
                continue
            s = self.at(child)
            if buf[s:s + 4] != b"SNOD":
                raise ValueError("bad symbol table node")
            nsym = struct.unpack_from("<H", buf, s + 6)[0]
            for k in range(nsym):
                no, oh = struct.unpack_from("<QQ", buf, s + 8 + 40 * k)
                e = buf.index(b"\0", heap_seg + no)
                members[bytes(buf[heap_seg + no:e]).decode()] = oh

    # ---- datasets --------------------------------------------------------------------------------
    def dataset(self, addr, name):
        dims = dtype = layout = None
        filters = []
        mclass = None
        is_group = False
        for mt, _mf, d in self.messages(addr):
            if mt == 0x01:
                ver, rank, fl = d[0], d[1], d[2]
                off = 8 if ver == 1 else 4
                dims = struct.unpack_from("<%dQ" % rank, d, off) if rank else ()
            elif mt == 0x03:
                cls, size = d[0] & 0x0F, struct.unpack_from("<I", d, 4)[0]
                if d[1] & 1:
                    raise ValueError(f"{name}: big-endian data are not supported")
                if cls == 1 and size in (4, 8):
                    dtype = np.dtype("<f%d" % size)
                elif cls == 0 and size in (1, 2, 4, 8):
                    dtype = np.dtype("<%s%d" % ("i" if d[1] & 0x08 else "u", size))
                else:
                    raise ValueError(f"{name}: HDF5 datatype class {cls} / size {size} is not supported (cell, struct, reference or string data)")
            elif mt == 0x08:
                layout = d
            elif mt == 0x0B:
                ver, nf = d[0], d[1]
                q = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid = struct.unpack_from("<H", d, q)[0]
                    if ver == 1 or fid >= 256:
                        nlen, _fl, ncv = struct.unpack_from("<HHH", d, q + 2)
                        q += 8 + (-(-nlen // 8) * 8 if ver == 1 else nlen)
                    else:
                        _fl, ncv = struct.unpack_from("<HH", d, q + 2)
                        q += 6
                    q += 4 * ncv + (4 if (ver == 1 and ncv % 2) else 0)
                    filters.append(fid)
            elif mt == 0x0C:
                ver = d[0]
                nsz, tsz, ssz = struct.unpack_from("<HHH", d, 2)
                q = 8 + (1 if ver == 3 else 0)
                pad = (lambda n: -(-n // 8) * 8) if ver == 1 else (lambda n: n)
                an = d[q:q + nsz].split(b"\0")[0].decode()
                q += pad(nsz)
                tcls, tsize = d[q] & 0x0F, struct.unpack_from("<I", d, q + 4)[0]
                q += pad(tsz) + pad(ssz)
                if an == "MATLAB_class" and tcls == 3:
                    mclass = d[q:q + tsize].split(b"\0")[0].decode()
            elif mt in (0x11, 0x02, 0x06):
                is_group = True
        if is_group or dims is None or dtype is None or layout is None:
            raise ValueError(f"{name}: not a plain numeric dataset (MATLAB cell / struct / object variables are not supported)")
        if mclass in ("cell", "struct", "function_handle"):
            raise ValueError(f"{name}: MATLAB class '{mclass}' is not supported")
        n = int(np.prod(dims)) if dims else 1
        raw = self._read_layout(layout, dims, dtype, filters, name, n)
        a = np.frombuffer(raw, dtype=dtype, count=n)
        a = a.reshape(dims[::-1], order="F") if dims else a.reshape(())  # HDF5 (n, m) row-major == MATLAB m x n column-major
        if mclass == "logical":
            a = a.astype(bool)
        elif mclass in (None, "double", "single") and a.dtype.kind == "f":
            a = a.astype(np.float64)
        return np.array(a)

    def _read_layout(self, d, dims, dtype, filters, name, n):
        ver, cls = d[0], d[1]
        if ver != 3:
            raise ValueError(f"{name}: data layout message version {ver} is not supported")
        nbytes = n * dtype.itemsize
        if cls == 0:  # compact
            sz = struct.unpack_from("<H", d, 2)[0]
            return bytes(d[4:4 + sz])
        if cls == 1:  # contiguous
            addr, sz = struct.unpack_from("<QQ", d, 2)
            if addr == UNDEF:
                return b"\0" * nbytes
            return bytes(self.buf[self.at(addr):self.at(addr) + nbytes])
        if cls != 2:
            raise ValueError(f"{name}: layout class {cls} is not supported")
        rank1 = d[2]
        bt = struct.unpack_from("<Q", d, 3)[0]
        cdims = struct.unpack_from("<%dI" % rank1, d, 11)
        rank = rank1 - 1
        if rank != len(dims):
            raise ValueError(f"{name}: chunk rank mismatch")
        out = np.zeros(dims, dtype=dtype)
        if bt != UNDEF:
            self._walk_chunk_btree(bt, rank, cdims, dtype, filters, out, name)
        return out.tobytes()

    def _walk_chunk_btree(self, addr, rank, cdims, dtype, filters, out, name):
        buf, p = self.buf, self.at(addr)
        if buf[p:p + 4] != b"TREE":
            raise ValueError(f"{name}: bad chunk B-tree node")
        ntype, level, used = struct.unpack_from("<BBH", buf, p + 4)
        ksz = 8 + 8 * (rank + 1)
        q = p + 24
        for i in range(used):
            csize, fmask = struct.unpack_from("<II", buf, q)
            offs = struct.unpack_from("<%dQ" % (rank + 1), buf, q + 8)
            child = struct.unpack_from("<Q", buf, q + ksz)[0]
            q += ksz + 8
            if level > 0:
                self._walk_chunk_btree(child, rank, cdims, dtype, filters, out, name)
                continue
            raw = bytes(buf[self.at(child):self.at(child) + csize])
            for k, fid in reversed(list(enumerate(filters))):
                if fmask & (1 << k):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:  # shuffle: byte planes -> elements
                    es = dtype.itemsize
                    ne = len(raw) // es
                    raw = np.frombuffer(raw, np.uint8, ne * es).reshape(es, ne).T.tobytes() + raw[ne * es:]
                else:
                    raise ValueError(f"{name}: HDF5 filter {fid} is not supported")
            chunk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(cdims[:rank]))).reshape(cdims[:rank])
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs[:rank], cdims[:rank], out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]


def load_mat73(path):
    """{name: array} of the numeric variables of a MATLAB -v7.3 file (real double / single / integer / logical matrices)."""
    with open(path, "rb") as f:
        buf = f.read()
    hf = _File(buf)
    out = {}
    for name, addr in hf.group_members(hf.root_hdr).items():
        if name.startswith("#"):  # "#refs#", "#subsystem#": storage of cell / struct / object variables
            continue
        out[name] = hf.dataset(addr, name)
    return out


def is_mat73(path):
    with open(path, "rb") as f:
        head = f.read(128)
    return head.startswith(b"MATLAB 7.3 MAT-file")
