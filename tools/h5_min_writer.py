"""Writer-format generator for the random-action episode file: emits the on-disk structures h5py's defaults produce
(environment/libero/lb_data/lb_randsam.py:84-104 calls `h5py.File(path, 'w')`, `create_group`, `create_dataset(name, data=ndarray)`,
`group.attrs[...] = ...` with no libver / chunks / compression arguments => HDF5 "earliest" format):

    superblock version 0 (offsets / lengths 8 bytes, group leaf K = 4, internal K = 16)
    groups       = version-1 object header with a symbol-table message -> v1 B-tree (TREE) of symbol-table nodes (SNOD, <= 8 entries,
                   names sorted) + local heap (HEAP) holding the names; further B-tree levels when a group has > 8 * 32 members
    datasets     = version-1 object header: dataspace v1, datatype (fixed / floating point, little endian), fill value, data layout
                   v3 contiguous (or chunked without filters, `chunks=`), modification time
    attributes   = scalar int64 / fixed-length string attribute messages on groups (env_seed, env_list_name)

h5py is not installed in the build image, so tests write their files with this module and read them with the native reader
(csrc/h5read.hip); the reader is ALSO checked against a file written by the HDF5 library itself (tests/golden/hdf5lib_sample.mat).
Written from the public HDF5 File Format Specification 2.0; no h5py / libhdf5 code involved.
"""
import struct
import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


class _File:
    def __init__(self):
        self.buf = bytearray(b"\0" * 96)          # superblock (56) + root symbol table entry (40), filled in at the end

    def alloc(self, data: bytes) -> int:
        self.buf += b"\0" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def patch(self, addr, data):
        self.buf[addr:addr + len(data)] = data


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(messages):
    body = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def _datatype(dt: np.dtype):
    dt = np.dtype(dt)
    if dt.kind in "ui":
        bits = 0x08 if dt.kind == "i" else 0x00                    # bit 3: signed; bit 0 = 0: little endian
        return struct.pack("<BBBBI", 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "f":
        if dt.itemsize == 4:
            props = struct.pack("<HHBBBBII", 0, 32, 23, 8, 0, 23, 127, 0)[:12]
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            sign = 31
        else:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            sign = 63
        return struct.pack("<BBBBI", 0x10 | 1, 0x20, sign, 0, dt.itemsize) + props
    raise TypeError(dt)


def _dataspace(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _attr(name: str, value):
    nm = _pad8(name.encode() + b"\0")
    if isinstance(value, (int, np.integer)):
        dt, sp, data = _datatype(np.int64), struct.pack("<BBB5x", 1, 0, 0), struct.pack("<q", int(value))
    else:
        raw = str(value).encode() + b"\0"
        dt = struct.pack("<BBBBI", 0x10 | 3, 0, 0, 0, len(raw))     # class 3 string, null-terminated, ASCII
        sp, data = struct.pack("<BBB5x", 1, 0, 0), raw
    dtp, spp = _pad8(dt), _pad8(sp)
    return _msg(0x0C, struct.pack("<BxHHH", 1, len(name) + 1, len(dt), len(sp)) + nm + dtp + spp + data)


def _write_dataset(f: _File, arr: np.ndarray, chunks=None):
    arr = np.ascontiguousarray(arr)
    msgs = [_msg(0x01, _dataspace(arr.shape)), _msg(0x03, _datatype(arr.dtype), flags=1),
            _msg(0x05, struct.pack("<BBBB", 2, 2, 2, 0))]           # fill value v2: allocate late, write if set, undefined
    if chunks is None:
        data_addr = f.alloc(arr.tobytes()) if arr.size else UNDEF
        msgs.append(_msg(0x08, struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes)))
    else:
        chunks = tuple(int(c) for c in chunks)
        assert len(chunks) == arr.ndim
        grid = [range(0, s, c) for s, c in zip(arr.shape, chunks)]
        entries = []
        for idx in np.ndindex(*[len(g) for g in grid]):
            off = [g[i] for g, i in zip(grid, idx)]
            block = np.zeros(chunks, dtype=arr.dtype)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, chunks, arr.shape))
            block[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
            entries.append((off, f.alloc(block.tobytes())))
        csize = int(np.prod(chunks)) * arr.itemsize

        def key(off):
            return struct.pack("<II", csize, 0) + b"".join(struct.pack("<Q", o) for o in off) + struct.pack("<Q", 0)

        def node(level, items):                                       # items: [(first offset, child address)]
            body = b"".join(key(o) + struct.pack("<Q", a) for o, a in items)
            last = [s for s in arr.shape]
            body += key(last)
            cap = 2 * 32 * (8 + 8 * (arr.ndim + 1) + 8) + 8 + 8 * (arr.ndim + 1)
            body += b"\0" * max(0, cap - len(body))
            return f.alloc(b"TREE" + struct.pack("<BBHQQ", 1, level, len(items), UNDEF, UNDEF) + body)

        level, items = 0, entries
        while True:
            groups = [items[i:i + 64] for i in range(0, len(items), 64)]
            items = [(g[0][0], node(level, g)) for g in groups]
            if len(items) == 1:
                break
            level += 1
        msgs.append(_msg(0x08, struct.pack("<BBBQ", 3, 2, arr.ndim + 1, items[0][1]) +
                         b"".join(struct.pack("<I", c) for c in chunks) + struct.pack("<I", arr.itemsize)))
    msgs.append(_msg(0x12, struct.pack("<B3xI", 1, 1700000000)))      # modification time
    return f.alloc(_object_header(msgs))


def _write_group(f: _File, members: dict, attrs: dict):
    """members: name -> object header address.  Returns (object header address, btree address, heap address)."""
    names = sorted(members, key=lambda s: s.encode())
    heap = bytearray(b"\0" * 8)                                        # offset 0: the empty name (key 0 of every B-tree)
    offs = {}
    for n in names:
        offs[n] = len(heap)
        heap += _pad8(n.encode() + b"\0")
    free_off = len(heap)
    heap += struct.pack("<QQ", 1, 16) + b"\0" * 0                        # one free block: next = 1 (none), size 16
    heap_data = f.alloc(bytes(heap))
    heap_addr = f.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data))

    def snod(chunk):
        body = b"".join(struct.pack("<QQII16x", offs[n], members[n], 0, 0) for n in chunk)
        body += b"\0" * (40 * (2 * LEAF_K - len(chunk)))
        return f.alloc(b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk)) + body)

    def tree(level, items):                                            # items: [(largest name offset in the child, child address)]
        body = struct.pack("<Q", 0)
        for koff, addr in items:
            body += struct.pack("<QQ", addr, koff)
        body += b"\0" * (16 * (2 * INTERNAL_K - len(items)))
        return f.alloc(b"TREE" + struct.pack("<BBHQQ", 0, level, len(items), UNDEF, UNDEF) + body)

    leaves = [names[i:i + 2 * LEAF_K] for i in range(0, len(names), 2 * LEAF_K)] or [[]]
    items = [(offs[c[-1]] if c else 0, snod(c)) for c in leaves]
    level = 0
    while True:
        groups = [items[i:i + 2 * INTERNAL_K] for i in range(0, len(items), 2 * INTERNAL_K)]
        items = [(g[-1][0], tree(level, g)) for g in groups]
        if len(items) == 1:
            break
        level += 1
    btree = items[0][1]
    msgs = [_msg(0x11, struct.pack("<QQ", btree, heap_addr))] + [_attr(k, v) for k, v in attrs.items()]
    return f.alloc(_object_header(msgs)), btree, heap_addr


def write_h5(path, tree: dict, attrs=None, chunks=None):
    """tree: nested dict; leaves are numpy arrays (datasets), inner dicts are groups.  attrs: {group path: {name: int | str}}.
    chunks: {dataset path: chunk shape} for datasets to be stored chunked (unfiltered)."""
    attrs, chunks = attrs or {}, chunks or {}
    f = _File()

    def emit(node, prefix):
        members = {}
        for name, child in node.items():
            p = f"{prefix}/{name}" if prefix else name
            members[name] = emit(child, p)[0] if isinstance(child, dict) else _write_dataset(f, np.asarray(child), chunks.get(p))
        return _write_group(f, members, attrs.get(prefix, {}))

    root_oh, root_bt, root_heap = emit(tree, "")
    eof = len(f.buf)
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, root_oh, 1, 0) + struct.pack("<QQ", root_bt, root_heap)
    assert len(sb) == 96
    f.patch(0, sb)
    with open(path, "wb") as fh:
        fh.write(bytes(f.buf))


def write_randsam_file(path, episodes: dict, env_list_name="libero-8tk-65to72-v3", action_dtype=np.float64, chunked=False):
    """episodes: {task: [(imgs uint8 [T+1,H,W,3], acts [T,7], ee_poses [T+1,3] or None, env_seed)]} -> the layout of lb_randsam.py:84-104."""
    tree, attrs, chunks = {}, {}, {}
    for task, eps in episodes.items():
        tree[task] = {}
        for i, (imgs, acts, ee, seed) in enumerate(eps):
            imgs = np.asarray(imgs, np.uint8)
            g = {"agentview_image": imgs, "action": np.asarray(acts, action_dtype),
                 "ee_poses": np.zeros((len(imgs), 3), np.float64) if ee is None else np.asarray(ee, np.float64)}
            tree[task][str(i)] = g
            attrs[f"{task}/{i}"] = {"env_seed": int(seed), "env_list_name": env_list_name}
            if chunked:
                chunks[f"{task}/{i}/agentview_image"] = (min(16, len(imgs)),) + imgs.shape[1:]
    write_h5(path, tree, attrs, chunks)
