"""HDF5 without h5py: the subset of the file format the reference's logits files use, in pure Python (struct + zlib + numpy).

The reference stores test-set logits with h5py (test.py:20-31, ``store_append_h5``): one file, datasets keyed like
``logits/action_<modk>`` (a group ``logits`` holding a 2-D float32 dataset), ``compression='gzip', compression_opts=9,
chunks=True, maxshape=(None, C)``, grown batch by batch with ``resize``; challenge.py reads them back for ensembling.  h5py is
not part of this image's python, so the same files are written (and read) here directly:

* ``write(path, {key: array})``   -- a new file: version-0 superblock, old-style groups (symbol-table B-tree + local heap),
  version-1 object headers, chunked layout (version-3 message, version-1 chunk B-tree), deflate level 9, unlimited first
  dimension -- what libhdf5 itself writes under h5py's default ``libver='earliest'``;
* ``append(path, {key: array})``  -- the reference's ``store_append_h5``: creates the file / dataset, or grows the dataset
  along dimension 0 IN PLACE (new chunks and a new chunk index at the end of the file, dimension sizes patched in the header);
* ``read(path)`` / ``read(path, key)`` -- datasets of such files, including files h5py wrote (contiguous or chunked layout,
  deflate and shuffle filters, integer and IEEE float types, header continuation blocks).

tests/test_h5lite_cpu.py checks all three against h5py itself where an interpreter with h5py exists (/opt/conda in this image):
h5py reads what this writes (names, shape, maxshape, dtype, chunks, gzip-9, values), h5py appends to it the reference's way,
and this reads what h5py wrote.
"""
import mmap
import os
import struct
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
GROUP_LEAF_K, GROUP_INTERNAL_K, CHUNK_K = 4, 16, 32     # library defaults; the first two are recorded in the superblock
UNLIMITED = UNDEF


def _pad8(n: int) -> int:
    return (n + 7) & ~7


# ------------------------------------------------------------------------------------------------------------- writer

class _Image(object):
    """The bytes being added to a file from offset `base` on (a new file: base 0): append-only, 8-byte aligned allocations,
    addresses are file offsets."""

    def __init__(self, base: int = 0, initial: bytes = b""):
        self.base = base
        self.b = bytearray(initial)

    def end(self) -> int:
        return self.base + len(self.b)

    def alloc(self, n: int) -> int:
        at = _pad8(len(self.b))
        self.b.extend(b"\0" * (at - len(self.b) + n))
        return self.base + at

    def put(self, at: int, data: bytes):
        self.b[at - self.base:at - self.base + len(data)] = data

    def add(self, data: bytes) -> int:
        at = self.alloc(len(data))
        self.put(at, data)
        return at


def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
    body = body + b"\0" * (_pad8(len(body)) - len(body))
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(messages: List[bytes], spare: int = 0) -> bytes:
    """Version-1 object header; `spare` bytes of NIL message leave the library room to add messages in place."""
    if spare:
        messages = messages + [_message(0x0000, b"\0" * spare)]
    body = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def _dtype_message(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        raise ValueError("h5lite writes little-endian types")
    if dt.kind == "f" and dt.itemsize in (4, 8):
        exp_bits, man_bits = (8, 23) if dt.itemsize == 4 else (11, 52)
        bits = dt.itemsize * 8
        return struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, dt.itemsize) + \
            struct.pack("<HHBBBBI", 0, bits, man_bits, exp_bits, 0, man_bits, (1 << (exp_bits - 1)) - 1)
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + \
            struct.pack("<HH", 0, dt.itemsize * 8)
    raise ValueError(f"h5lite: dtype {dt} is not supported")


def _dataspace_message(shape: Tuple[int, ...], maxshape: Tuple[int, ...]) -> bytes:
    return struct.pack("<BBB5x", 1, len(shape), 1) + b"".join(struct.pack("<Q", d) for d in shape) + \
        b"".join(struct.pack("<Q", d) for d in maxshape)


def _chunk_key(nbytes: int, mask: int, offsets: Tuple[int, ...]) -> bytes:
    return struct.pack("<II", nbytes, mask) + b"".join(struct.pack("<Q", o) for o in offsets) + struct.pack("<Q", 0)


def _write_chunk_index(img: _Image, chunks: List[Tuple[Tuple[int, ...], int, int]], end_offsets: Tuple[int, ...]) -> int:
    """Version-1 B-tree (node type 1) over `chunks` = [(element offsets, address, stored bytes)], sorted; returns the root address."""
    rank = len(end_offsets)
    key_size = 8 + 8 * (rank + 1)
    node_size = 24 + (2 * CHUNK_K + 1) * key_size + 2 * CHUNK_K * 8
    level = 0
    entries = [(off, addr, nbytes) for off, addr, nbytes in chunks]     # (first key offsets, child address, first key bytes)
    if not entries:
        at = img.alloc(node_size)
        img.put(at, b"TREE" + struct.pack("<BBHQQ", 1, 0, 0, UNDEF, UNDEF) + _chunk_key(0, 0, end_offsets))
        return at
    while True:
        groups = [entries[i:i + 2 * CHUNK_K] for i in range(0, len(entries), 2 * CHUNK_K)]
        addrs = [img.alloc(node_size) for _ in groups]
        nxt = []
        for gi, grp in enumerate(groups):
            left = addrs[gi - 1] if gi else UNDEF
            right = addrs[gi + 1] if gi + 1 < len(groups) else UNDEF
            body = b"TREE" + struct.pack("<BBHQQ", 1, level, len(grp), left, right)
            for off, addr, nbytes in grp:
                body += _chunk_key(nbytes, 0, off) + struct.pack("<Q", addr)
            last = groups[gi + 1][0] if gi + 1 < len(groups) else None
            body += _chunk_key(0, 0, last[0] if last else end_offsets)
            img.put(addrs[gi], body)
            nxt.append((grp[0][0], addrs[gi], grp[0][2]))
        if len(groups) == 1:
            return addrs[0]
        entries, level = nxt, level + 1


def _chunk_grid(shape, chunk):
    counts = [(s + c - 1) // c for s, c in zip(shape, chunk)]
    idx = [0] * len(shape)
    if any(c == 0 for c in counts):
        return
    while True:
        yield tuple(i * c for i, c in zip(idx, chunk))
        d = len(shape) - 1
        while d >= 0:
            idx[d] += 1
            if idx[d] < counts[d]:
                break
            idx[d] = 0
            d -= 1
        if d < 0:
            return


def _store_chunks(img: _Image, arr: np.ndarray, chunk: Tuple[int, ...], level: int, row0: int = 0):
    """Compresses `arr` (whose first row is element row `row0` of the dataset, row0 a multiple of the chunk rows) chunk by chunk."""
    out = []
    for off in _chunk_grid(arr.shape, chunk):
        block = np.zeros(chunk, dtype=arr.dtype)
        sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, chunk, arr.shape))
        block[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
        data = zlib.compress(block.tobytes(), level)
        out.append(((off[0] + row0,) + tuple(off[1:]), img.add(data), len(data)))
    return out


def _guess_chunk(shape: Tuple[int, ...], itemsize: int) -> Tuple[int, ...]:
    """Whole rows (all trailing dimensions), as many as fit ~512 KiB, at most the rows of the first batch (but >= 1)."""
    row_bytes = itemsize * int(np.prod(shape[1:], dtype=np.int64)) if len(shape) > 1 else itemsize
    rows = max(1, min(max(shape[0], 1), (512 * 1024) // max(row_bytes, 1)))
    return (rows,) + tuple(max(s, 1) for s in shape[1:])


def _dataset_header(img: _Image, arr: np.ndarray, chunk, level: int) -> int:
    shape = arr.shape
    maxshape = (UNLIMITED,) + tuple(shape[1:])
    chunks = _store_chunks(img, arr, chunk, level)
    end = (_pad_rows(shape[0], chunk[0]),) + (0,) * (len(shape) - 1)
    btree = _write_chunk_index(img, chunks, end)
    layout = struct.pack("<BBBQ", 3, 2, len(shape) + 1, btree) + b"".join(struct.pack("<I", c) for c in chunk) + \
        struct.pack("<I", arr.dtype.itemsize)
    name = b"deflate\0"
    pipeline = struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 1, len(name), 0x0001, 1) + name + struct.pack("<II", level, 0)
    msgs = [_message(0x0001, _dataspace_message(shape, maxshape)),
            _message(0x0003, _dtype_message(arr.dtype), flags=1),
            _message(0x0005, struct.pack("<BBBB", 2, 3, 2, 0)),          # fill value: incremental allocation, write if set, default
            _message(0x000B, pipeline),
            _message(0x0008, layout)]
    return img.add(_object_header(msgs, spare=64))


def _pad_rows(n: int, c: int) -> int:
    return (n + c - 1) // c * c


def _write_group(img: _Image, children: Dict[str, int], child_is_group: Dict[str, Tuple[int, int]]) -> Tuple[int, int, int]:
    """Old-style group over `children` = {name: object header address}; returns (header address, B-tree address, heap address)."""
    names = sorted(children, key=lambda s: s.encode())
    if len(names) > 2 * GROUP_LEAF_K * 2 * GROUP_INTERNAL_K:
        raise ValueError("h5lite: too many links in one group")
    heap_data = bytearray(8)                         # offset 0: the empty name
    name_off = {}
    for n in names:
        name_off[n] = len(heap_data)
        enc = n.encode() + b"\0"
        heap_data.extend(enc + b"\0" * (_pad8(len(enc)) - len(enc)))
    data_at = img.add(bytes(heap_data))
    heap_at = img.add(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, data_at))     # free list: none (1 = H5HL_FREE_NULL)
    snods = [names[i:i + 2 * GROUP_LEAF_K] for i in range(0, len(names), 2 * GROUP_LEAF_K)] or [[]]
    snod_at = []
    for grp in snods:
        body = b"SNOD" + struct.pack("<BBH", 1, 0, len(grp))
        for n in grp:
            if n in child_is_group:
                bt, hp = child_is_group[n]
                body += struct.pack("<QQII", name_off[n], children[n], 1, 0) + struct.pack("<QQ", bt, hp)
            else:
                body += struct.pack("<QQII", name_off[n], children[n], 0, 0) + b"\0" * 16
        body += b"\0" * (8 + 2 * GROUP_LEAF_K * 40 - len(body))
        snod_at.append(img.add(body))
    node_size = 24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8
    used = len(snods) if names else 0
    body = b"TREE" + struct.pack("<BBHQQ", 0, 0, used, UNDEF, UNDEF) + struct.pack("<Q", 0)
    for grp, at in zip(snods, snod_at):
        if names:
            body += struct.pack("<QQ", at, name_off[grp[-1]])
    body += b"\0" * (node_size - len(body))
    btree_at = img.add(body)
    hdr_at = img.add(_object_header([_message(0x0011, struct.pack("<QQ", btree_at, heap_at))], spare=40))
    return hdr_at, btree_at, heap_at


def _superblock(root_hdr: int, root_btree: int, root_heap: int, eof: int) -> bytes:
    return SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", GROUP_LEAF_K, GROUP_INTERNAL_K, 0) + \
        struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) + struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_btree, root_heap)


def write(path: str, datasets: Dict[str, np.ndarray], compression_level: int = 9,
          chunks: Optional[Dict[str, Tuple[int, ...]]] = None):
    """A new HDF5 file holding `datasets` = {'group/.../name': array}: gzip-compressed, chunked, first dimension unlimited."""
    img = _Image(0, b"\0" * 96)
    tree: Dict = {}
    for key, arr in datasets.items():
        parts = [p for p in key.split("/") if p]
        if not parts:
            raise ValueError("h5lite: empty dataset name")
        node = tree
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise ValueError(f"h5lite: {key}: a dataset is in the way")
        if np.ndim(arr) == 0:
            raise ValueError("h5lite: scalar datasets are not supported")
        arr = np.ascontiguousarray(arr)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        ch = tuple((chunks or {}).get(key) or _guess_chunk(arr.shape, arr.dtype.itemsize))
        node[parts[-1]] = ("dataset", arr, ch)

    def build(node):
        children, groups = {}, {}
        for name, v in node.items():
            if isinstance(v, dict):
                hdr, bt, hp = build(v)
                children[name], groups[name] = hdr, (bt, hp)
            else:
                children[name] = _dataset_header(img, v[1], v[2], compression_level)
        return _write_group(img, children, groups)

    root_hdr, root_bt, root_hp = build(tree)
    eof = _pad8(len(img.b))
    img.b.extend(b"\0" * (eof - len(img.b)))
    img.put(0, _superblock(root_hdr, root_bt, root_hp, eof))
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(bytes(img.b))
    os.replace(tmp, path)


# ------------------------------------------------------------------------------------------------------------- reader

class _Dataset(object):
    __slots__ = ("shape", "maxshape", "dtype", "layout", "chunk", "btree", "data_addr", "data_size", "filters",
                 "space_at", "layout_at", "header_at")


class File(object):
    """Read access to an HDF5 file of the subset described in the module docstring."""

    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self.size = os.fstat(fh.fileno()).st_size
            self.b = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) if self.size else b""
        if self.b[:8] != SIGNATURE:
            raise ValueError(f"{path}: not an HDF5 file (no signature at offset 0)")
        ver = self.b[8]
        if ver not in (0, 1) or self.b[13] != 8 or self.b[14] != 8:
            raise ValueError(f"{path}: superblock version {ver} / offset size {self.b[13]} is not supported (libver='earliest' files only)")
        self.leaf_k, self.internal_k = struct.unpack_from("<HH", self.b, 16)
        at = 24 + (4 if ver == 1 else 0)
        self.eof_at = at + 16
        self.base, _, self.eof, _ = struct.unpack_from("<QQQQ", self.b, at)
        _, self.root_header, _, _ = struct.unpack_from("<QQII", self.b, at + 32)
        self.objects: Dict[str, _Dataset] = {}
        self._walk(self.root_header, "")

    # -- object headers
    def _messages(self, at: int):
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", self.b, at)
        if ver != 1:
            raise ValueError("h5lite: object header version %d is not supported (libver='earliest' files only)" % ver)
        blocks = [(at + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", self.b, p)
                body_at = p + 8
                if mtype == 0x0010:
                    blocks.append(struct.unpack_from("<QQ", self.b, body_at))
                out.append((mtype, body_at, msize))
                p = body_at + msize
        return out

    def _walk(self, header_at: int, prefix: str):
        msgs = self._messages(header_at)
        st = [m for m in msgs if m[0] == 0x0011]
        if st:
            btree, heap = struct.unpack_from("<QQ", self.b, st[0][1])
            for name, child in self._group_entries(btree, heap):
                self._walk(child, prefix + "/" + name if prefix else name)
            return
        if any(m[0] in (0x0002, 0x000A) for m in msgs):
            raise ValueError("h5lite: new-style groups (link messages) are not supported (libver='earliest' files only)")
        if any(m[0] == 0x0008 for m in msgs):
            self.objects[prefix] = self._dataset(header_at, msgs)

    def _group_entries(self, btree: int, heap: int):
        if self.b[heap:heap + 4] != b"HEAP":
            raise ValueError("h5lite: bad local heap")
        data_at = struct.unpack_from("<Q", self.b, heap + 24)[0]
        out = []

        def name_at(off):
            p = data_at + off
            return self.b[p:self.b.find(b"\0", p)].decode()

        def node(at):
            if self.b[at:at + 4] == b"SNOD":
                n = struct.unpack_from("<H", self.b, at + 6)[0]
                for i in range(n):
                    noff, hdr = struct.unpack_from("<QQ", self.b, at + 8 + 40 * i)
                    out.append((name_at(noff), hdr))
                return
            if self.b[at:at + 4] != b"TREE":
                raise ValueError("h5lite: bad group B-tree node")
            _, _, used = struct.unpack_from("<BBH", self.b, at + 4)
            for i in range(used):
                node(struct.unpack_from("<Q", self.b, at + 24 + 8 + 16 * i)[0])

        node(btree)
        return out

    def _dataset(self, header_at: int, msgs) -> _Dataset:
        d = _Dataset()
        d.header_at, d.filters, d.maxshape = header_at, [], None
        for mtype, at, size in msgs:
            if mtype == 0x0001:
                ver, rank, flags = struct.unpack_from("<BBB", self.b, at)
                p = at + (8 if ver == 1 else 4)
                d.space_at = p
                d.shape = struct.unpack_from("<%dQ" % rank, self.b, p)
                d.maxshape = struct.unpack_from("<%dQ" % rank, self.b, p + 8 * rank) if flags & 1 else d.shape
            elif mtype == 0x0003:
                cv, b0, b1, _, sz = struct.unpack_from("<BBBBI", self.b, at)
                cls, order = cv & 15, ">" if b0 & 1 else "<"
                if cls == 1:
                    d.dtype = np.dtype(f"{order}f{sz}")
                elif cls == 0:
                    d.dtype = np.dtype(f"{order}{'i' if b0 & 8 else 'u'}{sz}")
                else:
                    raise ValueError(f"h5lite: datatype class {cls} is not supported")
            elif mtype == 0x0008:
                ver, cls = struct.unpack_from("<BB", self.b, at)
                if ver != 3:
                    raise ValueError(f"h5lite: data layout message version {ver} is not supported")
                d.layout, d.layout_at = cls, at
                if cls == 2:
                    ndim = self.b[at + 2]
                    d.btree = struct.unpack_from("<Q", self.b, at + 3)[0]
                    d.chunk = struct.unpack_from("<%dI" % ndim, self.b, at + 11)[:-1]
                elif cls == 1:
                    d.data_addr, d.data_size = struct.unpack_from("<QQ", self.b, at + 2)
                elif cls == 0:
                    d.data_size = struct.unpack_from("<H", self.b, at + 2)[0]
                    d.data_addr = at + 4
            elif mtype == 0x000B:
                ver, nf = struct.unpack_from("<BB", self.b, at)
                p = at + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid = struct.unpack_from("<H", self.b, p)[0]
                    if ver == 1 or fid >= 256:
                        nlen, flags, ncd = struct.unpack_from("<HHH", self.b, p + 2)
                        p += 8 + (_pad8(nlen) if ver == 1 else nlen)
                    else:
                        flags, ncd = struct.unpack_from("<HH", self.b, p + 2)
                        p += 6
                    cd = struct.unpack_from("<%dI" % ncd, self.b, p)
                    p += 4 * ncd + (4 if ver == 1 and ncd % 2 else 0)
                    d.filters.append((fid, cd))
        return d

    # -- data
    def _chunks(self, d: _Dataset):
        rank = len(d.shape)
        key_size = 8 + 8 * (rank + 1)
        out = []

        def node(at):
            if at == UNDEF:
                return
            if self.b[at:at + 4] != b"TREE":
                raise ValueError("h5lite: bad chunk B-tree node")
            ntype, level, used = struct.unpack_from("<BBH", self.b, at + 4)
            p = at + 24
            for _ in range(used):
                nbytes, mask = struct.unpack_from("<II", self.b, p)
                offs = struct.unpack_from("<%dQ" % rank, self.b, p + 8)
                child = struct.unpack_from("<Q", self.b, p + key_size)[0]
                if level:
                    node(child)
                else:
                    out.append((offs, child, nbytes, mask))
                p += key_size + 8

        node(d.btree)
        return out

    def keys(self):
        return list(self.objects)

    def info(self, key: str) -> Dict:
        d = self.objects[key.strip("/")]
        comp = [f for f in d.filters if f[0] == 1]
        return {"shape": tuple(d.shape), "maxshape": tuple(None if m == UNLIMITED else m for m in d.maxshape), "dtype": d.dtype,
                "chunks": tuple(d.chunk) if d.layout == 2 else None, "compression": "gzip" if comp else None,
                "compression_opts": comp[0][1][0] if comp else None}

    def close(self):
        if not isinstance(self.b, bytes):
            self.b.close()

    def __getitem__(self, key: str) -> np.ndarray:
        return self.rows(key, 0)

    def rows(self, key: str, row0: int) -> np.ndarray:
        """Rows row0.. of a dataset (only the chunks that hold them are decoded)."""
        d = self.objects[key.strip("/")]
        n = int(np.prod(d.shape, dtype=np.int64))
        if d.layout != 2:
            if d.data_addr == UNDEF or n == 0:
                return np.zeros(d.shape, d.dtype)[row0:]
            return np.frombuffer(self.b, d.dtype, n, d.data_addr).reshape(d.shape)[row0:].copy()
        out = np.zeros((max(d.shape[0] - row0, 0),) + tuple(d.shape[1:]), d.dtype)
        for offs, addr, nbytes, mask in self._chunks(d):
            if offs[0] + d.chunk[0] <= row0:
                continue
            raw = self.b[addr:addr + nbytes]
            for i, (fid, cd) in reversed(list(enumerate(d.filters))):
                if mask & (1 << i):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:          # shuffle: bytes of the elements were transposed
                    es = d.dtype.itemsize
                    raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                else:
                    raise ValueError(f"h5lite: filter {fid} is not supported")
            block = np.frombuffer(raw, d.dtype, int(np.prod(d.chunk, dtype=np.int64))).reshape(d.chunk)
            lo = max(offs[0], row0)                      # first dataset row of this chunk that is wanted
            sl = (slice(lo, min(offs[0] + d.chunk[0], d.shape[0])),) + \
                tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs[1:], d.chunk[1:], d.shape[1:]))
            if any(x.start >= x.stop for x in sl):
                continue
            src = (slice(lo - offs[0], sl[0].stop - offs[0]),) + tuple(slice(0, x.stop - x.start) for x in sl[1:])
            out[(slice(lo - row0, sl[0].stop - row0),) + sl[1:]] = block[src]
        return out


def read(path: str, key: Optional[str] = None):
    f = File(path)
    try:
        if key is not None:
            return f[key]
        return {k: f[k] for k in f.keys()}
    finally:
        f.close()


# ------------------------------------------------------------------------------------------------------------- append

def append(path: str, datasets: Dict[str, np.ndarray], compression_level: int = 9):
    """The reference's ``store_append_h5`` (test.py:20-31): create the file / the datasets, or grow existing datasets along
    dimension 0 by the given rows.  Growing happens in place: the rows go into new chunks at the end of the file (the last,
    partly filled chunk is re-written there), a new chunk index replaces the old one, and the dimension sizes, the index address
    and the end-of-file address are patched; what they replace stays behind as unreferenced bytes, which the format allows."""
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        return write(path, datasets, compression_level)
    f = File(path)
    try:
        missing = [k for k in datasets if k.strip("/") not in f.objects]
        if missing:        # a new dataset in an existing file: rebuild the file (the link tables would have to grow)
            everything = {k: f[k] for k in f.keys()}
            chunks = {k: f.info(k)["chunks"] for k in f.keys() if f.info(k)["chunks"]}
            for k, v in datasets.items():
                kk = k.strip("/")
                everything[kk] = np.concatenate([everything[kk], np.asarray(v, everything[kk].dtype)], 0) if kk in everything else np.asarray(v)
            f.close()
            return write(path, everything, compression_level, chunks)
        img = _Image(_pad8(f.size))
        patches = []
        for key, val in datasets.items():
            d = f.objects[key.strip("/")]
            val = np.ascontiguousarray(val, dtype=d.dtype.newbyteorder("<"))
            if d.layout != 2 or d.maxshape[0] != UNLIMITED or [fl[0] for fl in d.filters] != [1]:
                raise ValueError(f"h5lite.append: {key} is not a gzip-chunked dataset with an unlimited first dimension")
            if tuple(val.shape[1:]) != tuple(d.shape[1:]):
                raise ValueError(f"h5lite.append: {key}: rows of shape {val.shape[1:]} do not fit {tuple(d.shape[1:])}")
            if val.shape[0] == 0:
                continue
            old_rows, cr = d.shape[0], d.chunk[0]
            keep_rows = old_rows // cr * cr                       # rows in full chunks stay where they are
            kept = [(o, a, nb) for o, a, nb, _ in f._chunks(d) if o[0] < keep_rows]
            if keep_rows < old_rows:                              # the partly filled last chunk row: decode, extend, store again
                val = np.concatenate([f.rows(key, keep_rows), val], 0)
            new = _store_chunks(img, val, tuple(d.chunk), compression_level, row0=keep_rows)
            rows = keep_rows + val.shape[0]
            end = (_pad_rows(rows, cr),) + (0,) * (len(d.shape) - 1)
            btree = _write_chunk_index(img, sorted(kept + new), end)
            patches += [(d.space_at, struct.pack("<Q", rows)), (d.layout_at + 3, struct.pack("<Q", btree))]
        eof = _pad8(img.end())
        patches.append((f.eof_at, struct.pack("<Q", eof)))
        base = img.base
    finally:
        f.close()
    with open(path, "r+b") as fh:
        fh.seek(base)
        fh.write(bytes(img.b) + b"\0" * (eof - img.end()))
        for at, data in patches:
            fh.seek(at)
            fh.write(data)
