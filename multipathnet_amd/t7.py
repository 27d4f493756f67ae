"""Torch7 binary serialisation (`torch.save` / `torch.load`, default binary mode) — the wire format either side of the hot path:

  in   proposal tables        `torch.load(roidbfile)` -> {boxes = {FloatTensor[N,4], ...}, scores = {...}, images = {'x.jpg', ...}}
                              (DataSetJSON.lua:124-160)
  out  `boxes.t7`             aboxes[class][image] = FloatTensor[K,5]                        (run_test.lua:72-76)
       `results.t7` / saveResults tables of flat Float tensors                               (run_test.lua:79-83, utils.lua:335-372)

Format (torch7 File.lua writeObject / readObject, little-endian, 8-byte longs): every object starts with an int32 type tag —
0 nil, 1 number (f64), 2 string (int32 length + bytes), 3 table, 4 torch object, 5 boolean (int32).  Tables and torch
objects carry an int32 object index (shared references are written once; later occurrences repeat only the index).
  table         index, int32 n, n x (key object, value object)
  torch object  index, version string "V 1" (int32 length + bytes), class name (same), then the class's own payload:
    torch.XTensor   int32 nDim, int64 size[nDim], int64 stride[nDim], int64 storageOffset (1-based), storage object (or nil)
    torch.XStorage  int64 n, n raw elements
    tds.Hash        int64 n, n x (key object, value object)           tds.Vec   int64 n, n x value object
Python side: numbers -> float (ints that are whole stay float, as in Lua), strings -> str, tables -> dict (a table whose keys
are exactly 1..n -> list), tensors -> numpy arrays (C-contiguous), tds.Hash -> dict, tds.Vec -> list.  No Lua functions, no nn
modules (the pretrained model blobs are out of scope: weights enter through the C ABI as plain tensors).
"""
import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5

_DTYPES = {"Float": np.float32, "Double": np.float64, "Long": np.int64, "Int": np.int32, "Byte": np.uint8, "Char": np.int8, "Short": np.int16}
_NP2T = {np.dtype(v): k for k, v in _DTYPES.items()}


class T7Error(ValueError):
    pass


class TdsHash(dict):
    """tds.Hash (Tester_FRCNN.lua:58 builds img_boxes as one)"""


class TdsVec(list):
    """tds.Vec"""


class _Reader(object):
    def __init__(self, data):
        self.b, self.p, self.memo = data, 0, {}

    def _take(self, fmt):
        n = struct.calcsize(fmt)
        if self.p + n > len(self.b):
            raise T7Error("truncated .t7 stream at byte %d" % self.p)
        v = struct.unpack_from(fmt, self.b, self.p)
        self.p += n
        return v[0]

    def int(self):
        return self._take("<i")

    def long(self):
        return self._take("<q")

    def string(self):
        n = self.int()
        s = bytes(self.b[self.p:self.p + n])
        if len(s) != n:
            raise T7Error("truncated string")
        self.p += n
        return s.decode("latin-1")

    def obj(self):
        t = self.int()
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            return self._take("<d")
        if t == TYPE_STRING:
            return self.string()
        if t == TYPE_BOOLEAN:
            return self.int() != 0
        if t == TYPE_TABLE:
            idx = self.int()
            if idx in self.memo:
                return self.memo[idx]
            n = self.int()
            d = {}
            self.memo[idx] = d
            for _ in range(n):
                k = self.obj()
                d[k] = self.obj()
            lst = _as_list(d)
            if lst is not None:  # arrays come back as lists; (shared references to such a table are re-pointed)
                self.memo[idx] = lst
                return lst
            return d
        if t == TYPE_TORCH:
            idx = self.int()
            if idx in self.memo:
                return self.memo[idx]
            ver = self.string()
            cls = self.string() if ver.startswith("V ") else ver
            v = self._torch(cls, idx)
            self.memo[idx] = v
            return v
        raise T7Error("unsupported Torch7 type tag %d at byte %d (functions / userdata are not part of the path)" % (t, self.p - 4))

    def _torch(self, cls, idx):
        if cls.startswith("torch.") and cls.endswith("Storage"):
            dt = _DTYPES.get(cls[6:-7])
            if dt is None:
                raise T7Error("unsupported storage class " + cls)
            n = self.long()
            nb = n * np.dtype(dt).itemsize
            if n < 0 or self.p + nb > len(self.b):
                raise T7Error("truncated or corrupt %s: %d elements claimed at byte %d of %d" % (cls, n, self.p, len(self.b)))
            a = np.frombuffer(self.b, dtype=dt, count=n, offset=self.p).copy()
            self.p += nb
            return a
        if cls.startswith("torch.") and cls.endswith("Tensor"):
            dt = _DTYPES.get(cls[6:-6])
            if dt is None:
                raise T7Error("unsupported tensor class " + cls)
            nd = self.int()
            size = [self.long() for _ in range(nd)]
            stride = [self.long() for _ in range(nd)]
            off = self.long() - 1
            st = self.obj()
            if nd == 0 or st is None:
                return np.zeros((0,), dtype=dt) if nd == 0 else np.zeros(size, dtype=dt)
            it = np.dtype(dt).itemsize
            if not isinstance(st, np.ndarray) or st.dtype != np.dtype(dt):
                raise T7Error("%s does not reference a storage of its own type" % cls)
            # the view must stay inside its storage: sizes / strides / offset come unchecked from the file
            if off < 0 or any(n < 0 for n in size) or any(q < 0 for q in stride):
                raise T7Error("%s with negative size / stride / offset (size %s, stride %s, offset %d)" % (cls, size, stride, off + 1))
            if all(n >= 1 for n in size):
                last = off + sum((n - 1) * q for n, q in zip(size, stride))
                if last >= st.shape[0]:
                    raise T7Error("%s of size %s / stride %s / offset %d reaches element %d of a %d-element storage"
                                  % (cls, size, stride, off + 1, last + 1, st.shape[0]))
            else:
                return np.zeros(size, dtype=dt)
            v = np.lib.stride_tricks.as_strided(st[off:], shape=size, strides=[s * it for s in stride], writeable=False)
            return np.ascontiguousarray(v)
        if cls == "tds.Hash":
            n = self.long()
            h = TdsHash()
            for _ in range(n):
                k = self.obj()
                h[k] = self.obj()
            return h
        if cls == "tds.Vec":
            n = self.long()
            return TdsVec(self.obj() for _ in range(n))
        raise T7Error("unsupported torch class '%s' (only tensors, storages and tds containers travel on this path)" % cls)


def _as_list(d):
    n = len(d)
    if n == 0:
        return None
    for i in range(1, n + 1):
        if float(i) not in d:
            return None
    if any(not isinstance(k, float) for k in d):
        return None
    return [d[float(i)] for i in range(1, n + 1)]


class _Writer(object):
    def __init__(self):
        self.out, self.next_index, self.memo = [], 1, {}

    def int(self, v):
        self.out.append(struct.pack("<i", int(v)))

    def long(self, v):
        self.out.append(struct.pack("<q", int(v)))

    def string(self, s):
        b = s.encode("latin-1") if isinstance(s, str) else bytes(s)
        self.int(len(b))
        self.out.append(b)

    def _index(self, o):
        """(index, first occurrence?) — identity-keyed, like torch's objects table"""
        key = id(o)
        if key in self.memo:
            return self.memo[key][0], False
        idx = self.next_index
        self.next_index += 1
        self.memo[key] = (idx, o)  # keep `o` alive so that id() stays unique
        return idx, True

    def obj(self, o):
        if o is None:
            return self.int(TYPE_NIL)
        if isinstance(o, (bool, np.bool_)):
            self.int(TYPE_BOOLEAN)
            return self.int(1 if o else 0)
        if isinstance(o, (int, float, np.integer, np.floating)):
            self.int(TYPE_NUMBER)
            return self.out.append(struct.pack("<d", float(o)))
        if isinstance(o, (str, bytes)):
            self.int(TYPE_STRING)
            return self.string(o)
        if hasattr(o, "detach") and hasattr(o, "numpy"):  # a torch tensor (host side)
            o = o.detach().cpu().numpy()
        if isinstance(o, np.ndarray):
            return self._tensor(o)
        if isinstance(o, TdsHash):
            return self._tds(o, "tds.Hash")
        if isinstance(o, TdsVec):
            return self._tds(o, "tds.Vec")
        if isinstance(o, (list, tuple)):
            return self._table(o, [(float(i + 1), v) for i, v in enumerate(o)])
        if isinstance(o, dict):
            return self._table(o, [(float(k) if isinstance(k, (int, np.integer)) and not isinstance(k, bool) else k, v) for k, v in o.items()])
        raise T7Error("cannot serialise %r to Torch7" % type(o))

    def _table(self, o, items):
        self.int(TYPE_TABLE)
        idx, first = self._index(o)
        self.int(idx)
        if not first:
            return
        self.int(len(items))
        for k, v in items:
            self.obj(k)
            self.obj(v)

    def _header(self, o, cls):
        self.int(TYPE_TORCH)
        idx, first = self._index(o)
        self.int(idx)
        if first:
            self.string("V 1")
            self.string(cls)
        return first

    def _tds(self, o, cls):
        if not self._header(o, cls):
            return
        self.long(len(o))
        if cls == "tds.Hash":
            for k, v in o.items():
                self.obj(float(k) if isinstance(k, (int, np.integer)) else k)
                self.obj(v)
        else:
            for v in o:
                self.obj(v)

    def _tensor(self, a):
        name = _NP2T.get(a.dtype)
        if name is None:
            raise T7Error("no Torch7 tensor type for dtype %s" % a.dtype)
        if not self._header(a, "torch.%sTensor" % name):
            return
        c = np.ascontiguousarray(a)
        if c.ndim == 0:
            c = c.reshape(1)
        if c.size == 0:  # Torch7 has no zero-sized dimensions: resize2d(t, 0, 5) is the empty tensor — no dimensions, no storage
            self.int(0)
            self.long(1)
            return self.int(TYPE_NIL)
        self.int(c.ndim)
        for s in c.shape:
            self.long(s)
        for s in c.strides:
            self.long(s // c.itemsize)
        self.long(1)
        st = c.reshape(-1)
        self.int(TYPE_TORCH)  # its storage: a torch object of its own
        idx, _ = self._index(st)
        self.int(idx)
        self.string("V 1")
        self.string("torch.%sStorage" % name)
        self.long(st.size)
        self.out.append(st.tobytes())


def loads(data):
    """bytes of a Torch7 binary file -> Python objects (see the module docstring for the mapping)"""
    r = _Reader(memoryview(data))
    v = r.obj()
    return v


def load(path):
    with open(path, "rb") as f:
        return loads(f.read())


def dumps(obj):
    w = _Writer()
    w.obj(obj)
    return b"".join(w.out)


def save(path, obj):
    with open(path, "wb") as f:
        f.write(dumps(obj))
