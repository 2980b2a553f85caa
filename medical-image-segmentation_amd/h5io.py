"""Minimal pure-Python HDF5 reader and writer, enough for the model files Keras 2.x writes with
h5py's default ("earliest") file format and that the reference loads at
/root/reference utils/model.py:265 (`tf.keras.models.load_model(path, custom_objects=...)`):

  superblock version 0/1 (2/3 for the root address), "old style" groups (symbol table message,
  version-1 B-tree, local heap, symbol table nodes), version-1 object headers with continuation
  blocks (version-2 headers with compact link messages are read too), contiguous / compact /
  unfiltered chunked datasets of fixed-point, IEEE float and fixed-length string types, and
  attributes of those types plus variable-length strings (global heap).

h5py is not a dependency of this package and is not installed where it is developed; the reader
is pinned against an HDF5 file written by the HDF5 C library itself (see tests/test_h5io.py), the
writer against the reader.  Layout follows the "HDF5 File Format Specification Version 2.0".
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


# ------------------------------------------------------------------------------------------------
# datatypes
# ------------------------------------------------------------------------------------------------
class _Type:
    """Decoded datatype message: numpy dtype for fixed-size classes, or ('vlen_str',) marker."""

    def __init__(self, kind, size, dtype=None, base=None):
        self.kind, self.size, self.dtype, self.base = kind, size, dtype, base


def _parse_datatype(buf, off=0):
    cls_ver = buf[off]
    cls, ver = cls_ver & 0x0F, cls_ver >> 4
    bits = buf[off + 1] | (buf[off + 2] << 8) | (buf[off + 3] << 16)
    size = struct.unpack_from("<I", buf, off + 4)[0]
    if ver not in (1, 2, 3):
        raise H5Error("datatype message version %d" % ver)
    order = ">" if bits & 1 else "<"
    if cls == 0:                                   # fixed point
        signed = bool(bits & 0x08)
        return _Type("int", size, np.dtype("%s%s%d" % (order, "i" if signed else "u", size)))
    if cls == 1:                                   # IEEE float
        return _Type("float", size, np.dtype("%sf%d" % (order, size)))
    if cls == 3:                                   # fixed-length string
        return _Type("str", size, np.dtype("S%d" % size))
    if cls == 9:                                   # variable length
        if (bits & 0x0F) == 1:
            return _Type("vlen_str", size)
        base = _parse_datatype(buf, off + 8)
        return _Type("vlen_seq", size, base=base)
    if cls == 7:                                   # object reference
        return _Type("ref", size, np.dtype("<u%d" % size))
    raise H5Error("datatype class %d is not supported" % cls)


def _encode_datatype(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        # IEEE little-endian: sign location, exponent/mantissa geometry as libhdf5 writes them
        if dt.itemsize == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            bits = (0x20, 0x1F, 0x00)
        elif dt.itemsize == 8:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            bits = (0x20, 0x3F, 0x00)
        else:
            raise H5Error("float%d" % (8 * dt.itemsize))
        return bytes([0x11, bits[0], bits[1], bits[2]]) + struct.pack("<I", dt.itemsize) + props
    if dt.kind in "iu":
        b0 = 0x08 if dt.kind == "i" else 0x00
        return bytes([0x10, b0, 0, 0]) + struct.pack("<I", dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "S":
        return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", dt.itemsize)       # null-padded, ASCII
    raise H5Error("cannot encode dtype %r" % dt)


def _vlen_str_type(utf8):
    """Variable-length string (what h5py writes for Python str / bytes attribute values): class 9,
    type "string", character set UTF-8 or ASCII, parent type unsigned 8-bit."""
    return (bytes([0x19, 0x01, 0x01 if utf8 else 0x00, 0x00]) + struct.pack("<I", 16)
            + bytes([0x10, 0x00, 0, 0]) + struct.pack("<I", 1) + struct.pack("<HH", 0, 8))


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
class Dataset:
    def __init__(self, f, name, shape, typ, layout, attrs):
        self._f, self.name, self.shape, self._type, self._layout, self.attrs = f, name, shape, typ, layout, attrs

    @property
    def dtype(self):
        return self._type.dtype

    def read(self):
        return self._f._read_data(self.shape, self._type, self._layout)

    def __array__(self, dtype=None):
        a = self.read()
        return a.astype(dtype) if dtype is not None else a


class Group:
    def __init__(self, f, name, links, attrs):
        self._f, self.name, self._links, self.attrs = f, name, links, attrs

    def keys(self):
        return list(self._links)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            node = node._f._object(node._links[part], (node.name.rstrip("/") + "/" + part))
        return node

    def visit_datasets(self, prefix=""):
        for k in self.keys():
            o = self[k]
            if isinstance(o, Group):
                yield from o.visit_datasets(prefix + k + "/")
            else:
                yield prefix + k, o


class File(Group):
    """Read-only view of an HDF5 file held in memory."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        start = -1
        off = 0
        while off < len(self.buf):                # the superblock may sit after a user block
            if self.buf[off:off + 8] == SIGNATURE:
                start = off
                break
            off = 512 if off == 0 else off * 2
        if start < 0:
            raise H5Error("%s is not an HDF5 file" % path)
        b = self.buf
        ver = b[start + 8]
        if ver in (0, 1):
            self.O, self.L = b[start + 13], b[start + 14]
            p = start + 16 + 4 + 4 + (4 if ver == 1 else 0)
            self.base = self._uint(p, self.O)
            p += 4 * self.O
            root_header = self._uint(p + self.O, self.O)              # symbol table entry: name, header
        elif ver in (2, 3):
            self.O, self.L = b[start + 9], b[start + 10]
            p = start + 12
            self.base = self._uint(p, self.O)
            root_header = self._uint(p + 3 * self.O, self.O)
        else:
            raise H5Error("superblock version %d" % ver)
        if self.O != 8 or self.L != 8:
            raise H5Error("only 8-byte offsets and lengths are supported")
        if ver in (0, 1) and self.base == 0 and start != 0:
            self.base = start
        self._cache = {}
        root = self._object(root_header, "/")
        Group.__init__(self, self, "/", root._links, root.attrs)

    # ---- primitives -----------------------------------------------------------------------------
    def _uint(self, off, n):
        return int.from_bytes(self.buf[off:off + n], "little")

    def _abs(self, addr):
        return addr + self.base

    # ---- object headers -------------------------------------------------------------------------
    def _messages(self, addr):
        b = self.buf
        p = self._abs(addr)
        msgs = []
        if b[p:p + 4] == b"OHDR":
            flags = b[p + 5]
            q = p + 6
            if flags & 0x20:
                q += 16
            if flags & 0x10:
                q += 4
            nsz = 1 << (flags & 3)
            size = self._uint(q, nsz)
            q += nsz
            blocks = [(q, q + size)]
            track = bool(flags & 0x04)
            while blocks:
                q, end = blocks.pop(0)
                while q + 4 <= end - 0:
                    mtype, msize, mflags = b[q], struct.unpack_from("<H", b, q + 1)[0], b[q + 3]
                    q += 4 + (2 if track else 0)
                    if q + msize > end:
                        break
                    data = b[q:q + msize]
                    q += msize
                    if mtype == 0x10:
                        caddr, clen = struct.unpack_from("<QQ", data, 0)
                        ca = self._abs(caddr)
                        if b[ca:ca + 4] != b"OCHK":
                            raise H5Error("bad continuation chunk")
                        blocks.append((ca + 4, ca + clen - 4))
                    elif mtype != 0:
                        msgs.append((mtype, mflags, data))
                # the last 4 bytes of every chunk are its checksum (excluded from `end` for OCHK)
            return msgs
        if b[p] != 1:
            raise H5Error("object header version %d at %d" % (b[p], p))
        nmsg = struct.unpack_from("<H", b, p + 2)[0]
        hsize = struct.unpack_from("<I", b, p + 8)[0]
        blocks = [(p + 16, p + 16 + hsize)]
        while blocks and len(msgs) < nmsg + 64:
            q, end = blocks.pop(0)
            while q + 8 <= end:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, q)
                q += 8
                data = b[q:q + msize]
                q += msize
                if mtype == 0x10:
                    caddr, clen = struct.unpack_from("<QQ", data, 0)
                    blocks.append((self._abs(caddr), self._abs(caddr) + clen))
                elif mtype != 0:
                    msgs.append((mtype, mflags, data))
        return msgs

    def _object(self, addr, name):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        attrs = {}
        shape = typ = layout = None
        links = None
        filtered = False
        for mtype, mflags, d in msgs:
            if mtype == 0x01:
                shape = self._parse_dataspace(d)
            elif mtype == 0x03:
                typ = _parse_datatype(d)
            elif mtype == 0x08:
                layout = self._parse_layout(d)
            elif mtype == 0x0B:
                filtered = True
            elif mtype == 0x0C:
                k, v = self._parse_attribute(d)
                attrs[k] = v
            elif mtype == 0x11:
                btree, heap = struct.unpack_from("<QQ", d, 0)
                links = self._read_symbol_table(btree, heap)
            elif mtype == 0x06:
                links = links if links is not None else {}
                k, a = self._parse_link(d)
                if a is not None:
                    links[k] = a
            elif mtype == 0x02:
                links = links if links is not None else {}
                # link info: dense storage (fractal heap) is not supported
                flags = d[1]
                q = 2 + (8 if flags & 1 else 0)
                fheap = struct.unpack_from("<Q", d, q)[0]
                if fheap != UNDEF:
                    raise H5Error("group %s uses dense link storage, which this reader does not support" % name)
            elif mtype == 0x15:
                q = 2 + (2 if d[1] & 1 else 0)
                if struct.unpack_from("<Q", d, q)[0] != UNDEF:
                    raise H5Error("object %s uses dense attribute storage" % name)
        if links is not None and layout is None:
            obj = Group(self, name, links, attrs)
        elif layout is not None and typ is not None and shape is not None:
            if filtered and layout[0] == "chunked":
                raise H5Error("dataset %s is compressed/filtered, which this reader does not support" % name)
            obj = Dataset(self, name, shape, typ, layout, attrs)
        else:
            obj = Group(self, name, links or {}, attrs)
        self._cache[addr] = obj
        return obj

    # ---- groups -----------------------------------------------------------------------------------
    def _heap_string(self, heap_data_addr, off):
        p = self._abs(heap_data_addr) + off
        e = self.buf.index(b"\0", p)
        return self.buf[p:e].decode("utf-8")

    def _read_symbol_table(self, btree_addr, heap_addr):
        b = self.buf
        h = self._abs(heap_addr)
        if b[h:h + 4] != b"HEAP":
            raise H5Error("bad local heap signature")
        data_addr = struct.unpack_from("<Q", b, h + 8 + 2 * 8)[0]
        links = {}

        def walk(addr):
            p = self._abs(addr)
            if b[p:p + 4] == b"TREE":
                ntype, level, used = b[p + 4], b[p + 5], struct.unpack_from("<H", b, p + 6)[0]
                if ntype != 0:
                    raise H5Error("group B-tree expected")
                q = p + 8 + 16
                for i in range(used):
                    child = struct.unpack_from("<Q", b, q + 8)[0]       # key, child, key, child, ..., key
                    walk(child)
                    q += 16
            elif b[p:p + 4] == b"SNOD":
                n = struct.unpack_from("<H", b, p + 6)[0]
                q = p + 8
                for i in range(n):
                    name_off, header = struct.unpack_from("<QQ", b, q)
                    links[self._heap_string(data_addr, name_off)] = header
                    q += 40
            else:
                raise H5Error("bad group node signature %r" % b[p:p + 4])
        walk(btree_addr)
        return links

    def _parse_link(self, d):
        ver, flags = d[0], d[1]
        q = 2
        ltype = 0
        if flags & 0x08:
            ltype = d[q]
            q += 1
        if flags & 0x04:
            q += 8
        if flags & 0x10:
            q += 1
        nsz = 1 << (flags & 3)
        n = int.from_bytes(d[q:q + nsz], "little")
        q += nsz
        name = d[q:q + n].decode("utf-8")
        q += n
        if ltype != 0:
            return name, None                      # soft / external links are ignored
        return name, struct.unpack_from("<Q", d, q)[0]

    # ---- dataspace / layout / data ------------------------------------------------------------------
    def _parse_dataspace(self, d):
        ver, rank, flags = d[0], d[1], d[2]
        if ver == 1:
            q = 8
        elif ver == 2:
            q = 4
            if d[3] == 2:
                return None                        # null dataspace
        else:
            raise H5Error("dataspace version %d" % ver)
        return tuple(struct.unpack_from("<Q", d, q + 8 * i)[0] for i in range(rank))

    def _parse_layout(self, d):
        ver = d[0]
        if ver == 3:
            cls = d[1]
            if cls == 0:
                size = struct.unpack_from("<H", d, 2)[0]
                return ("compact", bytes(d[4:4 + size]))
            if cls == 1:
                addr, size = struct.unpack_from("<QQ", d, 2)
                return ("contiguous", addr, size)
            if cls == 2:
                nd = d[2]
                addr = struct.unpack_from("<Q", d, 3)[0]
                dims = struct.unpack_from("<%dI" % nd, d, 11)
                return ("chunked", addr, dims)
            raise H5Error("layout class %d" % cls)
        if ver in (1, 2):
            nd, cls = d[1], d[2]
            q = 8
            addr = None
            if cls != 0:
                addr = struct.unpack_from("<Q", d, q)[0]
                q += 8
            dims = struct.unpack_from("<%dI" % nd, d, q)
            q += 4 * nd
            if cls == 0:
                size = struct.unpack_from("<I", d, q)[0]
                return ("compact", bytes(d[q + 4:q + 4 + size]))
            if cls == 1:
                return ("contiguous", addr, None)
            return ("chunked", addr, dims)
        raise H5Error("data layout version %d" % ver)

    def _read_data(self, shape, typ, layout):
        count = int(np.prod(shape)) if shape else 1
        if typ.kind in ("vlen_str", "vlen_seq"):
            raw = self._raw(layout, count * 16, shape, 16)
            return self._vlen(raw, typ, shape)
        nbytes = count * typ.size
        raw = self._raw(layout, nbytes, shape, typ.size)
        a = np.frombuffer(raw, dtype=typ.dtype, count=count)
        return a.reshape(shape) if shape else a.reshape(())

    def _raw(self, layout, nbytes, shape, esize):
        if layout[0] == "compact":
            return layout[1][:nbytes]
        if layout[0] == "contiguous":
            if layout[1] == UNDEF:
                return bytes(nbytes)               # never written: fill value 0
            p = self._abs(layout[1])
            return self.buf[p:p + nbytes]
        # chunked, no filters: version-1 B-tree of raw data chunks
        addr, cdims = layout[1], layout[2]
        rank = len(shape)
        cshape = tuple(cdims[:rank])
        out = np.zeros(shape, dtype=np.dtype("V%d" % esize))
        b = self.buf

        def walk(a):
            p = self._abs(a)
            if b[p:p + 4] != b"TREE" or b[p + 4] != 1:
                raise H5Error("bad chunk B-tree node")
            level, used = b[p + 5], struct.unpack_from("<H", b, p + 6)[0]
            q = p + 24
            ksz = 8 + 8 * (rank + 1)
            for i in range(used):
                csize, fmask = struct.unpack_from("<II", b, q)
                offs = struct.unpack_from("<%dQ" % (rank + 1), b, q + 8)[:rank]
                child = struct.unpack_from("<Q", b, q + ksz)[0]
                if level > 0:
                    walk(child)
                else:
                    if fmask:
                        raise H5Error("filtered chunk")
                    cp = self._abs(child)
                    chunk = np.frombuffer(b[cp:cp + csize], dtype=out.dtype, count=int(np.prod(cshape))).reshape(cshape)
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, shape))
                    out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
                q += ksz + 8
        if addr != UNDEF:
            walk(addr)
        return out.tobytes()

    def _vlen(self, raw, typ, shape):
        out = []
        for i in range(len(raw) // 16):
            length, coll, idx = struct.unpack_from("<IQI", raw, 16 * i)
            data = self._global_heap_object(coll, idx) if length else b""
            if typ.kind == "vlen_str":
                out.append(data[:length].decode("utf-8"))
            else:
                out.append(np.frombuffer(data, dtype=typ.base.dtype, count=length))
        if not shape:
            return out[0]
        a = np.empty(len(out), dtype=object)
        a[:] = out
        return a.reshape(shape)

    def _global_heap_object(self, coll, idx):
        b = self.buf
        p = self._abs(coll)
        if b[p:p + 4] != b"GCOL":
            raise H5Error("bad global heap signature")
        size = struct.unpack_from("<Q", b, p + 8)[0]
        q, end = p + 16, p + size
        while q + 16 <= end:
            oidx, _ref, _res, osize = struct.unpack_from("<HHIQ", b, q)
            if oidx == idx:
                return b[q + 16:q + 16 + osize]
            if oidx == 0:
                break
            q += 16 + (osize + 7) // 8 * 8
        raise H5Error("global heap object %d not found" % idx)

    # ---- attributes ---------------------------------------------------------------------------------
    def _parse_attribute(self, d):
        ver = d[0]
        nsize, tsize, ssize = struct.unpack_from("<HHH", d, 2)
        q = 8
        if ver == 3:
            q += 1
        pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
        name = d[q:q + nsize].split(b"\0")[0].decode("utf-8")
        q += pad(nsize)
        typ = _parse_datatype(d, q)
        q += pad(tsize)
        shape = self._parse_dataspace(d[q:q + ssize])
        q += pad(ssize)
        if shape is None:
            return name, None
        count = int(np.prod(shape)) if shape else 1
        if typ.kind in ("vlen_str", "vlen_seq"):
            return name, self._vlen(d[q:q + 16 * count], typ, shape)
        a = np.frombuffer(d[q:q + count * typ.size], dtype=typ.dtype, count=count)
        if not shape:
            v = a[0]
            return name, (bytes(v).split(b"\0")[0] if typ.kind == "str" else v)
        return name, a.reshape(shape)


# ------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------
class _WGroup:
    def __init__(self):
        self.children = {}     # name -> _WGroup | np.ndarray
        self.attrs = {}


class Writer:
    """Builds a tree of groups, float/int/bytes datasets and attributes in memory and serialises it
    as an HDF5 file with the structures libhdf5 1.8 writes by default (superblock 0, symbol-table
    groups, version-1 object headers, contiguous datasets).  Every group is one symbol-table node
    under a one-entry B-tree; the node size (2K entries) is declared in the superblock."""

    def __init__(self):
        self.root = _WGroup()

    def group(self, path):
        g = self.root
        for part in [p for p in path.split("/") if p]:
            nxt = g.children.get(part)
            if nxt is None:
                nxt = g.children[part] = _WGroup()
            if not isinstance(nxt, _WGroup):
                raise H5Error("%s is a dataset" % path)
            g = nxt
        return g

    def dataset(self, path, array):
        parts = [p for p in path.split("/") if p]
        self.group("/".join(parts[:-1])).children[parts[-1]] = np.ascontiguousarray(array)

    def attr(self, path, name, value):
        self.group(path).attrs[name] = value

    # ---- serialisation ------------------------------------------------------------------------------
    def save(self, path):
        def count(g):
            n = len(g.children)
            for c in g.children.values():
                if isinstance(c, _WGroup):
                    n = max(n, count(c))
            return n
        leaf_k = max(4, (count(self.root) + 1) // 2)
        if leaf_k > 32767:
            raise H5Error("too many entries in one group")
        self.leaf_k = leaf_k
        self.out = bytearray(96)                       # superblock v0 with its root entry
        self.gheap = []                                # (placeholder offsets, bytes) of vlen strings
        root_addr, root_btree, root_heap = self._write_group(self.root)
        self._flush_global_heap()
        eof = len(self.out)
        sb = bytearray()
        sb += SIGNATURE
        sb += bytes([0, 0, 0, 0, 0, 8, 8, 0])
        sb += struct.pack("<HH", leaf_k, 16)
        sb += struct.pack("<I", 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_addr, 1, 0) + struct.pack("<QQ", root_btree, root_heap)
        self.out[:len(sb)] = sb
        tmp = str(path) + ".tmp"
        with open(tmp, "wb") as fh:
            fh.write(self.out)
        import os
        os.replace(tmp, str(path))

    def _alloc(self, data, align=8):
        while len(self.out) % align:
            self.out.append(0)
        addr = len(self.out)
        self.out += data
        return addr

    def _dataspace(self, shape):
        return bytes([1, len(shape), 0, 0, 0, 0, 0, 0]) + b"".join(struct.pack("<Q", s) for s in shape)

    def _attr_message(self, name, value):
        nm = name.encode("utf-8") + b"\0"
        fix = None
        if isinstance(value, (str, bytes)) and not isinstance(value, np.bytes_):
            typ, space = _vlen_str_type(isinstance(value, str)), self._dataspace(())
            data = bytearray(16)
            fix = value.encode("utf-8") if isinstance(value, str) else value
        else:
            a = np.asarray(value)
            if a.dtype.kind == "U":
                a = np.char.encode(a, "utf-8")
            if a.dtype.kind == "O":
                raise H5Error("attribute %s: object arrays are not supported" % name)
            typ, space = _encode_datatype(a.dtype), self._dataspace(a.shape)
            data = a.tobytes()
        p8 = lambda b: bytes(b) + bytes(-len(b) % 8)     # noqa: E731
        body = bytes([1, 0]) + struct.pack("<HHH", len(nm), len(typ), len(space)) + p8(nm) + p8(typ) + p8(space)
        return body, bytes(data), fix

    def _object_header(self, messages):
        """messages: list of (type, bytes); returns the address.  Attribute messages that carry a
        variable-length string register a fix-up for the global heap reference."""
        body = bytearray()
        fixups = []
        for mtype, data, fix in messages:
            data = bytes(data) + bytes(-len(data) % 8)
            body += struct.pack("<HHB3x", mtype, len(data), 0)
            if fix is not None:
                fixups.append((len(body) + fix[0], fix[1]))
            body += data
        if len(body) > 0xFFFF0000:
            raise H5Error("object header too large")
        head = struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body))
        addr = self._alloc(head + body)
        for off, payload in fixups:
            self.gheap.append((addr + 16 + off, payload))
        return addr

    def _attr_messages(self, attrs):
        msgs = []
        for k, v in attrs.items():
            body, data, fix = self._attr_message(k, v)
            if len(body) + len(data) > 64000:
                raise H5Error("attribute %s is larger than an object header message can hold; "
                              "split it (Keras does: layer_names0, layer_names1, ...)" % k)
            msgs.append((0x0C, body + data, (len(body), fix) if fix is not None else None))
        return msgs

    def _write_group(self, g):
        names = sorted(g.children, key=lambda s: s.encode("utf-8"))
        entries = []
        for n in names:
            c = g.children[n]
            if isinstance(c, _WGroup):
                addr, btree, heap = self._write_group(c)
                entries.append((n, addr, 1, struct.pack("<QQ", btree, heap)))
            else:
                entries.append((n, self._write_dataset(c), 0, bytes(16)))
        # local heap: offset 0 holds the empty string
        heap_data = bytearray(8)
        offs = []
        for n, *_ in entries:
            offs.append(len(heap_data))
            nb = n.encode("utf-8") + b"\0"
            heap_data += nb + bytes(-len(nb) % 8)
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)                         # one free block: next=1 (none), size
        data_addr = self._alloc(heap_data)
        heap_addr = self._alloc(b"HEAP" + bytes([0, 0, 0, 0]) + struct.pack("<QQQ", len(heap_data), free_off, data_addr))
        # symbol table node, padded to its declared capacity
        snod = bytearray(b"SNOD" + bytes([1, 0]) + struct.pack("<H", len(entries)))
        for (n, addr, ctype, scratch), off in zip(entries, offs):
            snod += struct.pack("<QQII", off, addr, ctype, 0) + scratch
        snod += bytes(40 * (2 * self.leaf_k - len(entries)))
        snod_addr = self._alloc(snod)
        # B-tree: a single leaf-level node with one child; sized for 2*16 children
        bt = bytearray(b"TREE" + bytes([0, 0]) + struct.pack("<H", 1 if entries else 0) + struct.pack("<QQ", UNDEF, UNDEF))
        bt += struct.pack("<Q", 0)
        if entries:
            bt += struct.pack("<QQ", snod_addr, offs[-1])
        bt += bytes(24 + (2 * 16) * 16 + 8 - len(bt))
        btree_addr = self._alloc(bt)
        msgs = [(0x11, struct.pack("<QQ", btree_addr, heap_addr), None)] + self._attr_messages(g.attrs)
        return self._object_header(msgs), btree_addr, heap_addr

    def _write_dataset(self, a):
        data_addr = self._alloc(a.tobytes()) if a.size else UNDEF
        msgs = [
            (0x01, self._dataspace(a.shape), None),
            (0x03, _encode_datatype(a.dtype), None),
            (0x05, bytes([2, 2, 0, 0]), None),                        # fill value v2: late alloc, never written
            (0x08, bytes([3, 1]) + struct.pack("<QQ", data_addr, a.nbytes), None),
        ]
        return self._object_header(msgs)

    def _flush_global_heap(self):
        if not self.gheap:
            return
        body = bytearray()
        refs = []
        for i, (where, payload) in enumerate(self.gheap):
            refs.append((where, len(payload), i + 1))
            body += struct.pack("<HHIQ", i + 1, 1, 0, len(payload)) + payload + bytes(-len(payload) % 8)
        size = 16 + len(body) + 16
        size = max(4096, (size + 7) // 8 * 8)
        free = size - 16 - len(body)
        body += struct.pack("<HHIQ", 0, 0, 0, free - 16) + bytes(free - 16)
        addr = self._alloc(b"GCOL" + bytes([1, 0, 0, 0]) + struct.pack("<Q", size) + bytes(body))
        for where, length, idx in refs:
            self.out[where:where + 16] = struct.pack("<IQI", length, addr, idx)
