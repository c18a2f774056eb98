"""A minimal pure-Python HDF5 subset: what a Keras-3 `model.weights.h5` needs, without h5py (not installable here).

STATUS: the CONTAINER is pinned against the real library, both ways.  h5py is not importable in the interpreter this package
runs in (and must not become a dependency); the image carries a second interpreter, /opt/conda/bin/python3.9, with h5py 3.3.0 on
libhdf5 1.10.6, which tests/test_against_second_interpreter.py uses in a subprocess as an independent implementation: real h5py opens
the files `write_file` produces and reads every dataset identically (values, dtype, shape incl. rank 0, empty groups), and
`read_file` returns exactly the tree real h5py wrote with its default `libver` (as Keras opens its weight files) -- more than 256
links in a group (two-level B-tree), attributes (continuation blocks), float16 / float64 / rank-0 int64 -- and refuses
libver="latest", chunked + gzip and big-endian files by name.  The byte layouts follow the published HDF5 File Format
Specification (version 1.x structures: the dialect libhdf5 writes with `libver="earliest"`).

Writer (`write_file`): superblock version 0, version-1 object headers, "old style" groups (symbol-table message, version-1
B-tree of one level + local heap + symbol-table nodes), contiguous little-endian float32 / float64 / int32 / int64 datasets,
no attributes, no chunking, no compression.
Reader (`read_file`): the same dialect, plus what real files of that dialect may contain beyond it -- object-header
continuation blocks, B-trees of any depth, version-2 dataspaces, compact layouts, float16.  Anything else (superblock 2 / 3,
"OHDR" version-2 object headers, chunked or filtered datasets, variable-length or compound types) raises `Hdf5Unsupported`
with the construct named, rather than returning wrong data.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16                  # group B-tree: 2 * LEAF_K symbols per node, 2 * INTERNAL_K children per tree node
HEAP_FREE_NULL = 1                          # local heap "no free block" marker

MSG_DATASPACE, MSG_DATATYPE, MSG_FILL, MSG_LAYOUT, MSG_CONTINUATION, MSG_SYMTAB = 0x0001, 0x0003, 0x0005, 0x0008, 0x0010, 0x0011


class Hdf5Unsupported(RuntimeError):
    pass


def _pad8(n: int) -> int:
    return (n + 7) & ~7


# --------------------------------------------------------------------------------------------------------------- writer
_DTYPES = {np.dtype("<f4"): (1, 4), np.dtype("<f8"): (1, 8), np.dtype("<i4"): (0, 4), np.dtype("<i8"): (0, 8)}


def _datatype_message(dt: np.dtype) -> bytes:
    cls, size = _DTYPES[dt]
    if cls == 1:        # IEEE float, little endian: mantissa normalisation "implied leading 1" (bits 4-5 = 2), sign in the top bit
        exp_size, mant = (8, 23) if size == 4 else (11, 52)
        head = bytes([0x11, 0x20, size * 8 - 1, 0x00]) + struct.pack("<I", size)
        props = struct.pack("<HHBBBBI", 0, size * 8, mant, exp_size, 0, mant, (1 << (exp_size - 1)) - 1)
    else:               # fixed point, little endian, two's complement (bit 3 of the class bits)
        head = bytes([0x10, 0x08, 0x00, 0x00]) + struct.pack("<I", size)
        props = struct.pack("<HH", 0, size * 8)
    return head + props


def _message(mtype: int, data: bytes) -> bytes:
    data = data + b"\0" * (_pad8(len(data)) - len(data))
    return struct.pack("<HHB3x", mtype, len(data), 0) + data


def _object_header(messages: List[bytes]) -> bytes:
    body = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)              # the superblock is filled in last

    def alloc(self, data: bytes) -> int:
        addr = len(self.buf)
        self.buf += data
        self.buf += b"\0" * (_pad8(len(self.buf)) - len(self.buf))
        return addr

    def dataset(self, arr: np.ndarray) -> int:
        arr = np.asarray(arr, order="C")                                  # (np.ascontiguousarray would turn a 0-d scalar into shape (1,))
        dt = np.dtype(arr.dtype).newbyteorder("<")                       # stored little endian whatever the host order
        if dt not in _DTYPES:
            raise Hdf5Unsupported(f"dataset dtype {arr.dtype} (float32 / float64 / int32 / int64 only)")
        raw = arr.astype(dt, copy=False).tobytes()
        data_addr = self.alloc(raw) if raw else UNDEF
        space = struct.pack("<BBBB4x", 1, arr.ndim, 0, 0) + b"".join(struct.pack("<Q", d) for d in arr.shape)
        fill = bytes([2, 2, 0, 0])                                        # version 2, allocate late, write at allocation, undefined
        layout = struct.pack("<BBQQ", 3, 1, data_addr, len(raw))         # version 3, contiguous
        return self.alloc(_object_header([_message(MSG_DATASPACE, space), _message(MSG_DATATYPE, _datatype_message(dt)),
                                          _message(MSG_FILL, fill), _message(MSG_LAYOUT, layout)]))

    def group(self, tree: dict) -> Tuple[int, int, int]:
        """Writes a group (children first); returns (object header, B-tree, heap) addresses."""
        entries = []                                                      # (name, header address, cache type, scratch)
        for name in sorted(tree, key=lambda s: s.encode("utf-8")):       # B-tree order = byte order of the link names
            child = tree[name]
            if isinstance(child, dict):
                oh, bt, hp = self.group(child)
                entries.append((name, oh, 1, struct.pack("<QQ", bt, hp)))
            else:
                entries.append((name, self.dataset(child), 0, b"\0" * 16))
        if len(entries) > 2 * LEAF_K * 2 * INTERNAL_K:
            raise Hdf5Unsupported(f"a group with {len(entries)} links needs a two-level B-tree (this writer stops at {2 * LEAF_K * 2 * INTERNAL_K})")
        # local heap: offset 0 holds the empty string (key 0 of every tree), names follow, each padded to 8 bytes
        heap, offs = bytearray(8), {}
        for name, *_ in entries:
            offs[name] = len(heap)
            enc = name.encode("utf-8") + b"\0"
            heap += enc + b"\0" * (_pad8(len(enc)) - len(enc))
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), HEAP_FREE_NULL, heap_data))
        # symbol-table nodes of up to 2 * LEAF_K entries, in name order (an empty group: a tree node with no entry, as libhdf5)
        nodes = []
        for i in range(0, len(entries), 2 * LEAF_K):
            chunk = entries[i:i + 2 * LEAF_K]
            body = b"".join(struct.pack("<QQI4x", offs[n], oh, ct) + sc for n, oh, ct, sc in chunk)
            body += b"\0" * (40 * (2 * LEAF_K - len(chunk)))
            nodes.append((self.alloc(b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk)) + body), offs[chunk[-1][0]]))
        # one B-tree node: key[0] = "" < every name, key[i + 1] = the largest name of child i
        keys_children = struct.pack("<Q", 0) + b"".join(struct.pack("<QQ", addr, last) for addr, last in nodes)
        node_size = 24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8
        tree_node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(nodes), UNDEF, UNDEF) + keys_children
        btree_addr = self.alloc(tree_node + b"\0" * (node_size - len(tree_node)))
        oh_addr = self.alloc(_object_header([_message(MSG_SYMTAB, struct.pack("<QQ", btree_addr, heap_addr))]))
        return oh_addr, btree_addr, heap_addr

    def finish(self, root: Tuple[int, int, int]) -> bytes:
        oh, bt, hp = root
        sb = SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQI4xQQ", 0, oh, 1, bt, hp)                    # root group symbol-table entry
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_file(tree: dict) -> bytes:
    """tree: nested dicts (groups) whose leaves are numpy arrays (datasets).  Returns the file's bytes."""
    w = _Writer()
    return w.finish(w.group(tree))


# --------------------------------------------------------------------------------------------------------------- reader
class _Reader:
    def __init__(self, data: bytes):
        self.d = data
        self.visiting = set()                 # object headers on the current path (a link cycle must not recurse for ever)
        if len(data) < 96 or data[:8] != SIGNATURE:
            raise Hdf5Unsupported("not an HDF5 file (signature missing at offset 0; user blocks are not supported)")
        ver = data[8]
        if ver not in (0, 1):
            raise Hdf5Unsupported(f"superblock version {ver} (files written with libver='latest'); only versions 0 / 1 are read")
        if data[13] != 8 or data[14] != 8:
            raise Hdf5Unsupported(f"offset / length sizes {data[13]} / {data[14]} (8 / 8 only)")
        off = 24 + (4 if ver == 1 else 0)                                    # version 1 adds the indexed-storage K + 2 reserved bytes
        self.base = struct.unpack_from("<Q", data, off)[0]
        self.root_entry = off + 32

    def u(self, fmt: str, off: int):
        """Bounds-checked little-endian unpack: a truncated or corrupt file raises Hdf5Unsupported, never struct.error."""
        size = struct.calcsize("<" + fmt)
        if off < 0 or off + size > len(self.d):
            raise Hdf5Unsupported(f"address {off} (+{size}) lies outside the file of {len(self.d)} bytes: truncated or corrupt")
        return struct.unpack_from("<" + fmt, self.d, off)

    def span(self, off: int, size: int) -> bytes:
        if off < 0 or size < 0 or off + size > len(self.d):
            raise Hdf5Unsupported(f"bytes {off} .. {off + size} lie outside the file of {len(self.d)} bytes: truncated or corrupt")
        return self.d[off:off + size]

    def messages(self, addr: int):
        """(type, data offset, size) of every message of a version-1 object header, continuation blocks included."""
        addr += self.base
        if self.span(addr, 4) == b"OHDR":
            raise Hdf5Unsupported("version-2 object header (libver='latest' files)")
        ver, _, nmsg, _, hsize = self.u("BBHII", addr)
        if ver != 1:
            raise Hdf5Unsupported(f"object header version {ver}")
        blocks, out, nblocks = [(addr + 16, hsize)], [], 0
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            nblocks += 1
            if nblocks > 4096 or pos + size > len(self.d):
                raise Hdf5Unsupported("object header continuation chain is cyclic or leaves the file")
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = self.u("HHB", pos)
                if mtype == MSG_CONTINUATION:
                    coff, clen = self.u("QQ", pos + 8)
                    blocks.append((coff + self.base, clen))
                out.append((mtype, pos + 8, msize))
                pos += 8 + msize
        return out

    def group_links(self, btree: int, heap: int) -> Dict[str, Tuple[int, int, int, int]]:
        """name -> (object header address, cache type, scratch B-tree, scratch heap) of an old-style group."""
        heap += self.base
        if self.span(heap, 4) != b"HEAP":
            raise Hdf5Unsupported("local heap signature missing")
        seg = self.u("Q", heap + 24)[0] + self.base
        links: Dict[str, Tuple[int, int, int, int]] = {}
        seen_nodes = set()

        def name_at(off: int) -> str:
            start = seg + off
            if start < 0 or start >= len(self.d):
                raise Hdf5Unsupported("link name offset leaves the file")
            end = self.d.find(b"\0", start, start + 4096)
            if end < 0:
                raise Hdf5Unsupported("unterminated link name")
            return self.d[start:end].decode("utf-8", "replace")

        def walk(addr: int, depth: int = 0):
            addr += self.base
            if addr in seen_nodes or depth > 16:
                raise Hdf5Unsupported("group B-tree is cyclic or deeper than 16 levels")
            seen_nodes.add(addr)
            sig = self.span(addr, 4)
            if sig == b"TREE":
                ntype, _level, used = self.u("BBH", addr + 4)
                if ntype != 0:
                    raise Hdf5Unsupported("a chunk B-tree where a group B-tree is expected")
                for i in range(used):
                    walk(self.u("Q", addr + 24 + 8 + 16 * i)[0], depth + 1)  # key0 | child0 key1 | child1 key2 ...
            elif sig == b"SNOD":
                nsym = self.u("H", addr + 6)[0]
                for i in range(nsym):
                    e = addr + 8 + 40 * i
                    noff, oh, ctype = self.u("QQI", e)
                    bt, hp = self.u("QQ", e + 24)
                    links[name_at(noff)] = (oh, ctype, bt, hp)
            else:
                raise Hdf5Unsupported(f"unexpected node signature {sig!r} in a group B-tree")

        walk(btree)
        return links

    def dataset(self, addr: int) -> np.ndarray:
        shape = dtype = raw = None
        for mtype, pos, size in self.messages(addr):
            if mtype == MSG_DATASPACE:
                ver, rank, flags = self.u("BBB", pos)
                if ver == 1:
                    dims = pos + 8
                elif ver == 2:
                    dims = pos + 4
                else:
                    raise Hdf5Unsupported(f"dataspace version {ver}")
                shape = tuple(self.u("Q", dims + 8 * i)[0] for i in range(rank))
            elif mtype == MSG_DATATYPE:
                cv, bits0 = self.u("BB", pos)
                cls, dsize = cv & 0x0F, self.u("I", pos + 4)[0]
                if bits0 & 1:
                    raise Hdf5Unsupported("big-endian dataset")
                if cls == 1 and dsize in (2, 4, 8):
                    dtype = np.dtype(f"<f{dsize}")
                elif cls == 0 and dsize in (1, 2, 4, 8):
                    dtype = np.dtype(f"<{'i' if bits0 & 8 else 'u'}{dsize}")
                else:
                    raise Hdf5Unsupported(f"datatype class {cls} of {dsize} bytes (floats and integers only)")
            elif mtype == MSG_LAYOUT:
                ver, lclass = self.u("BB", pos)
                if ver != 3:
                    raise Hdf5Unsupported(f"data layout message version {ver}")
                if lclass == 1:
                    daddr, dlen = self.u("QQ", pos + 2)
                    raw = b"" if daddr == UNDEF else self.span(daddr + self.base, dlen)
                elif lclass == 0:
                    dlen = self.u("H", pos + 2)[0]
                    raw = self.span(pos + 4, dlen)
                else:
                    raise Hdf5Unsupported("chunked dataset (Keras writes its weights contiguous)")
            elif mtype == 0x000B:
                raise Hdf5Unsupported("filtered (compressed) dataset")
        if shape is None or dtype is None or raw is None:
            raise Hdf5Unsupported("dataset without dataspace / datatype / layout message")
        if len(shape) > 32 or any(d > 1 << 40 for d in shape):
            raise Hdf5Unsupported(f"implausible dataspace {shape[:4]}...")
        count = 1
        for d in shape:
            count *= int(d)
        if len(raw) < count * dtype.itemsize:
            raise Hdf5Unsupported("dataset shorter than its dataspace")
        return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape).copy()

    def node(self, oh: int, ctype: int, bt: int, hp: int):
        if oh in self.visiting or len(self.visiting) > 64:
            raise Hdf5Unsupported("links form a cycle (or groups nest deeper than 64)")
        self.visiting.add(oh)
        try:
            if ctype != 1:                       # not cached in the link: the object header says what it is
                sym = [(p, s) for t, p, s in self.messages(oh) if t == MSG_SYMTAB]
                if not sym:
                    return self.dataset(oh)
                bt, hp = self.u("QQ", sym[0][0])
            return {name: self.node(*link) for name, link in self.group_links(bt, hp).items()}
        finally:
            self.visiting.discard(oh)

    def root(self) -> dict:
        _noff, oh, ctype = self.u("QQI", self.root_entry)
        bt, hp = self.u("QQ", self.root_entry + 24)
        return self.node(oh, ctype, bt, hp)


def read_file(data: bytes) -> dict:
    """The file as nested dicts (groups) of numpy arrays (datasets).  Raises Hdf5Unsupported -- and nothing else -- for anything
    outside the subset and for truncated or corrupt input (every address is bounds-checked, B-trees and links are checked for
    cycles): a file from outside must not crash, hang or be misread."""
    try:
        return _Reader(bytes(data)).root()
    except Hdf5Unsupported:
        raise
    except (struct.error, ValueError, IndexError, OverflowError, RecursionError, MemoryError, UnicodeError) as exc:
        raise Hdf5Unsupported(f"corrupt HDF5 structure ({type(exc).__name__}: {exc})") from exc
