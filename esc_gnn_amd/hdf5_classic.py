"""
Reader for "classic" HDF5 files -- what h5py / libhdf5 write by default (libver='earliest'), i.e. the files the
reference's own writers produce (scripts/python/generate_ellipsoid_dataset.py:1185-1206 and
scripts/python/parallel_process_dataset.py:193-230: h5py.File(...,'w'), create_dataset(..., compression='gzip')):

    superblock version 0 / 1, optional user block (signature searched at 0, 512, 1024, ...)
    old-style groups: version-1 object header -> Symbol Table message -> version-1 B-tree (node type 0) over
        Symbol Table Nodes (SNOD), link names in a Local Heap (HEAP)
    datasets: version-1 object headers (with continuation blocks); Dataspace v1/v2, Datatype classes 0 (fixed point)
        and 1 (floating point), Data Layout v1/v2/v3 compact / contiguous / chunked (version-1 B-tree, node type 1),
        Filter Pipeline v1/v2 with deflate (zlib, filter 1) and shuffle (filter 2)

written from the HDF5 File Format Specification (version 3.0, sections II.A, III.A-III.D, IV.A).  Same interface as
hdf5_min.H5Reader (which reads the version-2 files this repo writes): f["P"]["0"]["indices"][...].

PARITY UNPINNED: neither h5py nor libhdf5 exists in the build or GPU images.  The group / object-header / contiguous
part is checked against a real libhdf5-written file that ships with scipy (a MATLAB v7.3 file, tests/test_hdf5_min.py);
the chunked + deflate part only against byte fixtures assembled from the specification by tests/h5_classic_writer.py.
"""
import struct
import zlib
import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class ClassicH5Error(ValueError):
    pass


class ClassicH5Reader:
    def __init__(self, path: str):
        self.buf = np.memmap(path, dtype=np.uint8, mode="r")
        n = self.buf.shape[0]
        off = 0
        while True:
            if off + 8 > n:
                raise ClassicH5Error("no HDF5 signature found")
            if bytes(self.buf[off:off + 8]) == _SIG:
                break
            off = 512 if off == 0 else off * 2
        ver = int(self.buf[off + 8])
        if ver not in (0, 1):
            raise ClassicH5Error(f"superblock version {ver}: not a classic file (version 2/3 files go to hdf5_min.H5Reader)")
        b = self._bytes(off, 24)
        self.O, self.L = b[13], b[14]                      # size of offsets / lengths
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise ClassicH5Error(f"unsupported offset / length sizes {self.O} / {self.L}")
        self.leaf_k, self.int_k = struct.unpack("<HH", b[16:20])
        p = off + 24 + (4 if ver == 1 else 0)
        self.base = self._int(p, self.O)
        p += 4 * self.O                                    # base, free-space info, end of file, driver info
        # root group symbol table entry
        self.root_ohdr = self._addr(p + self.O)
        self._nodes = {}

    # ---------------------------------------------------------------- raw access
    def _bytes(self, p, n):
        if p < 0 or p + n > self.buf.shape[0]:
            raise ClassicH5Error(f"read of {n} bytes at {p} is outside the file")
        return bytes(self.buf[p:p + n])

    def _int(self, p, n):
        return int.from_bytes(self._bytes(p, n), "little")

    def _addr(self, p):
        """an address field: relative to the base address; all ones = undefined"""
        v = self._int(p, self.O)
        return None if v == (1 << (8 * self.O)) - 1 else v + self.base

    # ---------------------------------------------------------------- object headers (version 1)
    def _messages(self, addr):
        b = self._bytes(addr, 16)
        if b[0] != 1:
            if b[:4] == b"OHDR":
                raise ClassicH5Error("version-2 object header inside a classic file (libver='latest'?): not supported here")
            raise ClassicH5Error(f"no version-1 object header at {addr}")
        nmsg = struct.unpack("<H", b[2:4])[0]
        hsize = struct.unpack("<I", b[8:12])[0]
        blocks = [(addr + 16, hsize)]                      # the first block starts 8-byte aligned after the 12-byte prefix
        out = []
        while blocks and len(out) < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, _ = struct.unpack("<HHB", self._bytes(p, 5))
                body = self._bytes(p + 8, msize)
                p += 8 + msize
                if mtype == 0x0010:                        # continuation: offset, length
                    blocks.append((int.from_bytes(body[:self.O], "little") + self.base,
                                   int.from_bytes(body[self.O:self.O + self.L], "little")))
                out.append((mtype, body))
        return out

    def node(self, addr):
        hit = self._nodes.get(addr)
        if hit is None:
            hit = self._nodes[addr] = self._parse_node(addr)
        return hit

    def _parse_node(self, addr):
        msgs = self._messages(addr)
        d = {}
        for t, body in msgs:
            d.setdefault(t, body)
        if 0x0011 in d:                                     # Symbol Table message: a group
            st = d[0x0011]
            btree = int.from_bytes(st[:self.O], "little") + self.base
            heap = int.from_bytes(st[self.O:2 * self.O], "little") + self.base
            return _ClassicGroup(self, btree, heap)
        if 0x0001 in d and 0x0003 in d and 0x0008 in d:
            return self._parse_dataset(d)
        raise ClassicH5Error(f"object at {addr} is neither an old-style group nor a dataset")

    # ---------------------------------------------------------------- groups
    def _heap_data(self, heap_addr):
        b = self._bytes(heap_addr, 8 + 2 * self.L + self.O)
        if b[:4] != b"HEAP":
            raise ClassicH5Error(f"no local heap at {heap_addr}")
        size = int.from_bytes(b[8:8 + self.L], "little")
        seg = int.from_bytes(b[8 + 2 * self.L:8 + 2 * self.L + self.O], "little") + self.base
        return self._bytes(seg, size)

    def _group_links(self, btree_addr, heap_addr):
        heap = self._heap_data(heap_addr)
        links = {}
        entry = 2 * self.O + 24
        stack = [btree_addr]
        while stack:
            a = stack.pop()
            sig = self._bytes(a, 4)
            if sig == b"TREE":
                ntype, level, used = struct.unpack("<BBH", self._bytes(a + 4, 4))
                if ntype != 0:
                    raise ClassicH5Error("group B-tree node of the wrong type")
                p = a + 8 + 2 * self.O                      # past left / right sibling
                for i in range(used):                       # key_i, child_i, ..., key_used
                    child = self._int(p + self.L + i * (self.L + self.O), self.O) + self.base
                    stack.append(child)
            elif sig == b"SNOD":
                nsym = struct.unpack("<H", self._bytes(a + 6, 2))[0]
                for i in range(nsym):
                    q = a + 8 + i * entry
                    name_off = self._int(q, self.O)
                    end = heap.index(b"\x00", name_off)
                    links[heap[name_off:end].decode()] = self._int(q + self.O, self.O) + self.base
            else:
                raise ClassicH5Error(f"neither TREE nor SNOD at {a}")
        return links

    # ---------------------------------------------------------------- datasets
    def _parse_dataset(self, d):
        sp, ty, lay = d[0x0001], d[0x0003], d[0x0008]
        sv, rank, sflags = sp[0], sp[1], sp[2]
        if sv == 1:
            q = 8
        elif sv == 2:
            q = 4
        else:
            raise ClassicH5Error(f"dataspace message version {sv}")
        shape = tuple(int.from_bytes(sp[q + self.L * i:q + self.L * (i + 1)], "little") for i in range(rank))
        cls, tsize = ty[0] & 0x0F, struct.unpack("<I", ty[4:8])[0]
        big = ty[1] & 1
        if cls == 0:
            kind = "i" if ty[1] & 0x08 else "u"
        elif cls == 1:
            kind = "f"
        else:
            raise ClassicH5Error(f"datatype class {cls} is not fixed- or floating-point")
        dtype = np.dtype((">" if big else "<") + kind + str(tsize))
        filters = self._parse_filters(d.get(0x000B))
        lv = lay[0]
        if lv == 3:
            lclass = lay[1]
            if lclass == 0:                                 # compact
                n = struct.unpack("<H", lay[2:4])[0]
                return _ClassicDataset(self, shape, dtype, "compact", lay[4:4 + n], None, filters)
            if lclass == 1:                                 # contiguous
                a = int.from_bytes(lay[2:2 + self.O], "little")
                n = int.from_bytes(lay[2 + self.O:2 + self.O + self.L], "little")
                addr = None if a == (1 << (8 * self.O)) - 1 else a + self.base
                return _ClassicDataset(self, shape, dtype, "contiguous", (addr, n), None, filters)
            if lclass == 2:                                 # chunked
                nd = lay[2]
                a = int.from_bytes(lay[3:3 + self.O], "little")
                dims = struct.unpack("<" + "I" * nd, lay[3 + self.O:3 + self.O + 4 * nd])
                addr = None if a == (1 << (8 * self.O)) - 1 else a + self.base
                return _ClassicDataset(self, shape, dtype, "chunked", addr, dims[:-1], filters)
            raise ClassicH5Error(f"layout class {lclass}")
        if lv in (1, 2):
            nd, lclass = lay[1], lay[2]
            q = 8
            addr = None
            if lclass != 0:
                a = int.from_bytes(lay[q:q + self.O], "little")
                addr = None if a == (1 << (8 * self.O)) - 1 else a + self.base
                q += self.O
            dims = struct.unpack("<" + "I" * nd, lay[q:q + 4 * nd])
            q += 4 * nd
            if lclass == 2:
                return _ClassicDataset(self, shape, dtype, "chunked", addr, dims[:-1], filters)
            if lclass == 1:
                return _ClassicDataset(self, shape, dtype, "contiguous", (addr, int(np.prod(shape)) * dtype.itemsize), None, filters)
            n = struct.unpack("<I", lay[q:q + 4])[0]
            return _ClassicDataset(self, shape, dtype, "compact", lay[q + 4:q + 4 + n], None, filters)
        raise ClassicH5Error(f"data layout message version {lv} (libver='latest' files are not classic files)")

    @staticmethod
    def _parse_filters(body):
        if body is None:
            return []
        ver, nf = body[0], body[1]
        out = []
        p = 8 if ver == 1 else 2
        for _ in range(nf):
            fid = struct.unpack("<H", body[p:p + 2])[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack("<H", body[p:p + 2])[0]
                p += 2
            flags, ncd = struct.unpack("<HH", body[p:p + 4])
            p += 4
            if ver == 1:
                nlen = (nlen + 7) & ~7
            p += nlen
            cd = struct.unpack("<" + "I" * ncd, body[p:p + 4 * ncd])
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, flags, cd))
        return out

    def _chunks(self, btree_addr, nd):
        """leaves of a version-1 raw-data-chunk B-tree: (offsets, address, stored size, filter mask)"""
        out = []
        keysz = 8 + 8 * (nd + 1)
        stack = [btree_addr]
        while stack:
            a = stack.pop()
            if self._bytes(a, 4) != b"TREE":
                raise ClassicH5Error(f"no B-tree node at {a}")
            ntype, level, used = struct.unpack("<BBH", self._bytes(a + 4, 4))
            if ntype != 1:
                raise ClassicH5Error("chunk B-tree node of the wrong type")
            p = a + 8 + 2 * self.O
            for i in range(used):
                k = self._bytes(p, keysz)
                child = self._int(p + keysz, self.O) + self.base
                p += keysz + self.O
                if level > 0:
                    stack.append(child)
                else:
                    csize, fmask = struct.unpack("<II", k[:8])
                    offs = struct.unpack("<" + "Q" * (nd + 1), k[8:])[:nd]
                    out.append((offs, child, csize, fmask))
        return out

    # ---------------------------------------------------------------- h5py-like surface
    def __getitem__(self, key):
        return self.node(self.root_ohdr)[key]

    def __contains__(self, key):
        return key in self.node(self.root_ohdr)

    def keys(self):
        return self.node(self.root_ohdr).keys()

    def close(self):
        del self.buf


class _ClassicGroup:
    def __init__(self, f, btree, heap):
        self.f, self._btree, self._heap, self._links = f, btree, heap, None

    @property
    def links(self):
        if self._links is None:
            self._links = self.f._group_links(self._btree, self._heap)
        return self._links

    def __getitem__(self, key):
        node = self
        for part in [p for p in key.split("/") if p]:
            node = node.f.node(node.links[part])
        return node

    def __contains__(self, key):
        return key in self.links

    def keys(self):
        return self.links.keys()


class _ClassicDataset:
    def __init__(self, f, shape, dtype, kind, where, chunk, filters):
        self.f, self.shape, self.dtype, self.kind, self.where, self.chunk, self.filters = f, shape, dtype, kind, where, chunk, filters

    def _unfilter(self, raw, fmask):
        for n, (fid, _, cd) in reversed(list(enumerate(self.filters))):
            if fmask & (1 << n):
                continue                                    # this filter was skipped for this chunk
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:                                  # shuffle: bytes of the elements were de-interleaved
                es = cd[0] if cd else self.dtype.itemsize
                k = len(raw) // es
                raw = np.frombuffer(raw[:k * es], dtype=np.uint8).reshape(es, k).T.tobytes() + raw[k * es:]
            else:
                raise ClassicH5Error(f"filter {fid} is not supported (deflate = 1 and shuffle = 2 are)")
        return raw

    def __getitem__(self, idx):
        n = int(np.prod(self.shape)) if self.shape else 1
        if self.kind == "compact":
            arr = np.frombuffer(self.where[:n * self.dtype.itemsize], dtype=self.dtype).reshape(self.shape)
        elif self.kind == "contiguous":
            addr, size = self.where
            if addr is None or n == 0:
                arr = np.zeros(self.shape, self.dtype)
            else:
                arr = np.frombuffer(self.f._bytes(addr, n * self.dtype.itemsize), dtype=self.dtype).reshape(self.shape)
        else:
            arr = np.zeros(self.shape, self.dtype)
            nd = len(self.shape)
            if self.where is not None and n:
                for offs, addr, csize, fmask in self.f._chunks(self.where, nd):
                    raw = self._unfilter(self.f._bytes(addr, csize), fmask)
                    blk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(self.chunk))).reshape(self.chunk)
                    sel_o = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, self.chunk, self.shape))
                    sel_c = tuple(slice(0, s.stop - s.start) for s in sel_o)
                    arr[sel_o] = blk[sel_c]
        if self.dtype.byteorder == ">":
            arr = arr.astype(self.dtype.newbyteorder("<"))
        return arr[idx]
