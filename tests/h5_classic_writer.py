"""
Test-side writer of "classic" HDF5 files (superblock v0, old-style groups, chunked + deflate datasets), assembled from
the HDF5 File Format Specification v3.0 -- the shapes h5py / libhdf5 give the reference's operator files
(generate_ellipsoid_dataset.py:1185-1206: create_dataset(..., compression='gzip'); `size` uncompressed).  Neither h5py
nor libhdf5 exists in this image, so esc_gnn_amd/hdf5_classic.py cannot be checked against a file the reference wrote;
this writer builds byte fixtures from the specification instead (sections II.A superblock, III.A B-trees, III.B/C
symbol table nodes, III.D local heaps, IV.A object headers and messages).  Test infrastructure only.

    write_classic(path, {"P/0/indices": array, ...}, chunked=lambda name, arr: bool, userblock=0)
"""
import struct
import zlib
import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INT_K = 4, 16            # libhdf5's defaults: a symbol node holds <= 2*4 entries, a B-tree node <= 2*16 children


class _File:
    def __init__(self):
        self.b = bytearray()

    def alloc(self, data: bytes, align=8) -> int:
        while len(self.b) % align:
            self.b.append(0)
        a = len(self.b)
        self.b += data
        return a


def _msg(mtype, body):
    body = body + b"\x00" * (-len(body) % 8)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _ohdr(msgs):
    body = b"".join(msgs)
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body


def _dtype_msg(dt: np.dtype):
    if dt.kind in "iu":
        bits = 0x08 if dt.kind == "i" else 0
        return struct.pack("<B3BI", 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    spec = {2: (15, 10, 5, 0, 10, 15), 4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023)}[dt.itemsize]
    sign, eloc, esize, mloc, msize, bias = spec
    return struct.pack("<B3BI", 0x10 | 1, 0x20, sign, 0, dt.itemsize) + \
        struct.pack("<HHBBBBI", 0, 8 * dt.itemsize, eloc, esize, mloc, msize, bias)


def _dataspace_msg(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", s) for s in shape)


def _chunk_btree(f, entries, nd, elem):
    """entries: [(offsets, addr, size)] sorted; returns the root node's address (one level, or two when needed)."""
    def node(level, items, last_key_offs):
        body = b"TREE" + struct.pack("<BBH", 1, level, len(items)) + struct.pack("<QQ", UNDEF, UNDEF)
        for offs, addr, size in items:
            body += struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in offs) + struct.pack("<Q", 0)
            body += struct.pack("<Q", addr)
        body += struct.pack("<II", 0, 0) + b"".join(struct.pack("<Q", o) for o in last_key_offs) + struct.pack("<Q", 0)
        return f.alloc(body)
    cap = 2 * INT_K
    end = tuple(0 for _ in range(nd))
    if len(entries) <= cap:
        return node(0, entries, end)
    kids = []
    for i in range(0, len(entries), cap):
        part = entries[i:i + cap]
        kids.append((part[0][0], node(0, part, end), part[0][2]))
    assert len(kids) <= cap
    return node(1, kids, end)


def _dataset(f, arr: np.ndarray, chunked: bool, chunk_rows=None):
    arr = np.ascontiguousarray(arr)
    msgs = [_msg(0x0001, _dataspace_msg(arr.shape)), _msg(0x0003, _dtype_msg(arr.dtype))]
    if not chunked or arr.size == 0:
        addr = f.alloc(arr.tobytes()) if arr.size else UNDEF
        msgs.append(_msg(0x0008, struct.pack("<BBQQ", 3, 1, addr, arr.nbytes)))
        return f.alloc(_ohdr(msgs))
    nd = arr.ndim
    # chunk shape: whole leading dims, the last dim cut (several chunks, the last one partial)
    cshape = list(arr.shape)
    cshape[-1] = chunk_rows or max(1, (arr.shape[-1] + 2) // 3)
    entries = []
    for o in range(0, arr.shape[-1], cshape[-1]):
        blk = np.zeros(cshape, arr.dtype)
        piece = arr[..., o:o + cshape[-1]]
        blk[..., :piece.shape[-1]] = piece
        raw = zlib.compress(blk.tobytes(), 4)
        offs = tuple([0] * (nd - 1) + [o])
        entries.append((offs, f.alloc(raw), len(raw)))
    root = _chunk_btree(f, entries, nd, arr.dtype.itemsize)
    lay = struct.pack("<BBB", 3, 2, nd + 1) + struct.pack("<Q", root) + \
        b"".join(struct.pack("<I", c) for c in cshape) + struct.pack("<I", arr.dtype.itemsize)
    msgs.append(_msg(0x0008, lay))
    # filter pipeline v1: deflate (id 1), name omitted, one client value (level), padded
    msgs.append(_msg(0x000B, struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<II", 4, 0)))
    return f.alloc(_ohdr(msgs))


def _group(f, children):
    """children: {name: object header address}; returns the group's object header address."""
    names = sorted(children)                      # libhdf5 keeps symbol nodes sorted by name
    heap = bytearray(b"\x00" * 8)                 # offset 0: the empty name
    offs = {}
    for nm in names:
        offs[nm] = len(heap)
        heap += nm.encode() + b"\x00"
        while len(heap) % 8:
            heap.append(0)
    heap_data = f.alloc(bytes(heap))
    heap_addr = f.alloc(b"HEAP" + struct.pack("<B3x", 0) + struct.pack("<QQQ", len(heap), UNDEF, heap_data))
    cap = 2 * LEAF_K
    snods = []
    for i in range(0, max(len(names), 1), cap):
        part = names[i:i + cap]
        body = b"SNOD" + struct.pack("<BBH", 1, 0, len(part))
        for nm in part:
            body += struct.pack("<QQII16x", offs[nm], children[nm], 0, 0)
        body += b"\x00" * (40 * (cap - len(part)))
        snods.append((offs[part[-1]] if part else 0, f.alloc(body)))

    def tree(level, items):
        body = b"TREE" + struct.pack("<BBH", 0, level, len(items)) + struct.pack("<QQ", UNDEF, UNDEF)
        body += struct.pack("<Q", 0)
        for key, addr in items:
            body += struct.pack("<Q", addr) + struct.pack("<Q", key)
        return f.alloc(body)
    level, items = 0, snods
    while len(items) > 2 * INT_K:
        nxt = []
        for i in range(0, len(items), 2 * INT_K):
            part = items[i:i + 2 * INT_K]
            nxt.append((part[-1][0], tree(level, part)))
        items, level = nxt, level + 1
    btree = tree(level, items)
    return f.alloc(_ohdr([_msg(0x0011, struct.pack("<QQ", btree, heap_addr))])), btree, heap_addr


def write_classic(path, datasets, chunked=lambda name, arr: not name.endswith("size"), userblock=0, chunk_rows=None):
    f = _File()
    f.b += b"\x00" * 96                          # superblock v0 (56 bytes) + root symbol table entry (40 bytes)
    tree = {}
    for name, arr in datasets.items():
        parts = name.strip("/").split("/")
        d = tree
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = np.asarray(arr)

    def build(d):
        kids = {}
        for k, v in d.items():
            kids[k] = build(v)[0] if isinstance(v, dict) else _dataset(f, v, chunked(k, v), chunk_rows)
        return _group(f, kids)
    root, btree, heap = build(tree)
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", LEAF_K, INT_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, len(f.b), UNDEF)
    sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap)
    f.b[:len(sb)] = sb
    with open(path, "wb") as fh:
        if userblock:
            # addresses stay relative to the base address = the user block's size
            sbu = bytearray(f.b)
            sbu[24:32] = struct.pack("<Q", userblock)
            fh.write(b"\x00" * userblock)
            fh.write(bytes(sbu))
        else:
            fh.write(bytes(f.b))
