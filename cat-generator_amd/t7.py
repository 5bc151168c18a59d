"""Torch7 binary serialisation: the on-disk format of `torch.save` / `torch.load`, which the reference
uses for its checkpoints (train.lua:252-261 writes {D, G, opt, plot_data, epoch, normalize_*}; train.lua:127-142 and
sample.lua:69-75 read them back).

The format is that of torch7's File.lua `writeObject` in binary mode, little endian, 8-byte longs:
    object    := int32 type, payload
    type 0    nil
    type 1    number   : float64
    type 2    string   : int32 length, bytes
    type 3    table    : int32 index, [first time only:] int32 count, count x (key object, value object)
    type 4    torch    : int32 index, [first time only:] string "V 1", string class name, class payload
    type 5    boolean  : int32 0 / 1
    tensor payload   : int32 nDim, nDim x int64 size, nDim x int64 stride, int64 storageOffset (1-based), storage object
    storage payload  : int64 count, count x element
    any other class  : one table object holding the instance's fields (File.lua's default for classes without write())
`index` numbers referenced objects from 1 in order of first appearance; a repeated index is a back-reference.

The torch7 sources are not part of /root/reference and no Torch7 runs in this image, so this restatement is pinned only
by its round trip and by a hand-assembled byte string of the layout above (tests/test_t7.py): PARITY UNPINNED against
a real torch.save file.  Everything here is host-side Python; nothing on the training path imports it."""
import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5

_TENSORS = {"torch.FloatTensor": np.float32, "torch.DoubleTensor": np.float64, "torch.LongTensor": np.int64,
            "torch.IntTensor": np.int32, "torch.ByteTensor": np.uint8, "torch.CudaTensor": np.float32}
_STORAGES = {k.replace("Tensor", "Storage"): v for k, v in _TENSORS.items()}
_BY_DTYPE = {np.dtype(np.float32): "Float", np.dtype(np.float64): "Double", np.dtype(np.int64): "Long",
             np.dtype(np.int32): "Int", np.dtype(np.uint8): "Byte"}


class TorchObject:
    """An instance of a torch class other than a tensor / storage: its class name and its field table."""

    def __init__(self, typename, fields=None):
        self.typename = typename
        self.fields = dict(fields or {})

    def __getitem__(self, k):
        return self.fields[k]

    def get(self, k, default=None):
        return self.fields.get(k, default)

    def __repr__(self):
        return f"<{self.typename} {sorted(map(str, self.fields))}>"


class Storage:
    """torch.<T>Storage: a flat array (LongStorage sizes of nn.View, nn.Concat ...)."""

    def __init__(self, data, kind=None):
        self.data = np.ascontiguousarray(data).reshape(-1)
        self.kind = kind or _BY_DTYPE[self.data.dtype]


class CudaTensor:
    """Marks an array to be written as torch.CudaTensor / torch.CudaStorage (what cutorch writes for device tensors)."""

    def __init__(self, array):
        self.array = np.ascontiguousarray(array, dtype=np.float32)


def array_table(seq):
    """A Lua array: keys 1..n."""
    return {i + 1: v for i, v in enumerate(seq)}


def table_list(t):
    """The array part of a table read back (keys 1..n in order)."""
    out, i = [], 1
    while i in t:
        out.append(t[i])
        i += 1
    return out


class Writer:
    def __init__(self, f):
        self.f = f
        self.index = {}     # id(python object) -> torch index
        self.keep = []      # keeps the written objects alive so that ids stay unique
        self.next = 1

    def _int(self, v):
        self.f.write(struct.pack("<i", int(v)))

    def _long(self, v):
        self.f.write(struct.pack("<q", int(v)))

    def _string(self, s):
        b = s if isinstance(s, bytes) else s.encode("latin-1")
        self._int(len(b))
        self.f.write(b)

    def _ref(self, obj):
        """Writes the index; True if the object was written before (nothing else to emit)."""
        k = id(obj)
        if k in self.index:
            self._int(self.index[k])
            return True
        self.index[k] = self.next
        self.keep.append(obj)
        self._int(self.next)
        self.next += 1
        return False

    def _header(self, typename):
        self._string("V 1")
        self._string(typename)

    def _storage(self, st, cuda=False):
        self._int(TYPE_TORCH)
        if self._ref(st):
            return
        self._header("torch.CudaStorage" if cuda else f"torch.{st.kind}Storage")
        self._long(st.data.size)
        self.f.write(st.data.tobytes())

    def _tensor(self, holder, a, cuda):
        self._int(TYPE_TORCH)
        if self._ref(holder):
            return
        self._header("torch.CudaTensor" if cuda else f"torch.{_BY_DTYPE[a.dtype]}Tensor")
        if a.size == 0:              # torch.Tensor(): no dimensions, no storage
            self._int(0)
            self._long(1)
            self._int(TYPE_NIL)
            return
        self._int(a.ndim)
        for d in a.shape:
            self._long(d)
        stride = 1
        strides = []
        for d in reversed(a.shape):
            strides.append(stride)
            stride *= d
        for s in reversed(strides):
            self._long(s)
        self._long(1)
        self._storage(Storage(a), cuda)

    def write(self, obj):
        if obj is None:
            self._int(TYPE_NIL)
        elif isinstance(obj, (bool, np.bool_)):
            self._int(TYPE_BOOLEAN)
            self._int(1 if obj else 0)
        elif isinstance(obj, (int, float, np.integer, np.floating)):
            self._int(TYPE_NUMBER)
            self.f.write(struct.pack("<d", float(obj)))
        elif isinstance(obj, (str, bytes)):
            self._int(TYPE_STRING)
            self._string(obj)
        elif isinstance(obj, np.ndarray):
            a = np.ascontiguousarray(obj)
            if a.dtype not in _BY_DTYPE:
                raise TypeError(f"no torch tensor type for dtype {a.dtype}")
            self._tensor(obj, a, False)
        elif isinstance(obj, CudaTensor):
            self._tensor(obj, obj.array, True)
        elif isinstance(obj, Storage):
            self._storage(obj)
        elif isinstance(obj, TorchObject):
            self._int(TYPE_TORCH)
            if self._ref(obj):
                return
            self._header(obj.typename)
            self.write(obj.fields)
        elif isinstance(obj, dict):
            self._int(TYPE_TABLE)
            if self._ref(obj):
                return
            items = [(k, v) for k, v in obj.items() if v is not None]   # a Lua table cannot hold nil
            self._int(len(items))
            for k, v in items:
                self.write(k)
                self.write(v)
        elif isinstance(obj, (list, tuple)):
            self.write(array_table(obj))
        else:
            raise TypeError(f"cannot serialise {type(obj).__name__}")


class Reader:
    def __init__(self, f):
        self.f = f
        self.objects = {}

    def _unpack(self, fmt, n):
        b = self.f.read(n)
        if len(b) != n:
            raise EOFError("truncated torch7 file")
        return struct.unpack(fmt, b)[0]

    def _int(self):
        return self._unpack("<i", 4)

    def _long(self):
        return self._unpack("<q", 8)

    def _string(self):
        n = self._int()
        return self.f.read(n).decode("latin-1")

    def read(self):
        t = self._int()
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            v = self._unpack("<d", 8)
            return int(v) if v.is_integer() and abs(v) < 2 ** 53 else v
        if t == TYPE_STRING:
            return self._string()
        if t == TYPE_BOOLEAN:
            return self._int() != 0
        if t == TYPE_TABLE:
            idx = self._int()
            if idx in self.objects:
                return self.objects[idx]
            out = self.objects[idx] = {}
            for _ in range(self._int()):
                k = self.read()
                out[k] = self.read()
            return out
        if t == TYPE_TORCH:
            idx = self._int()
            if idx in self.objects:
                return self.objects[idx]
            version = self._string()
            typename = self._string() if version.startswith("V ") else version   # pre-versioning files: the name comes first
            if typename in _STORAGES:
                n = self._long()
                dt = np.dtype(_STORAGES[typename])
                data = np.frombuffer(self.f.read(n * dt.itemsize), dtype=dt).copy()
                self.objects[idx] = data
                return data
            if typename in _TENSORS:
                nd = self._int()
                size = [self._long() for _ in range(nd)]
                stride = [self._long() for _ in range(nd)]
                off = self._long() - 1
                self.objects[idx] = None   # the storage below takes its own index
                storage = self.read()
                if storage is None or nd == 0:
                    a = np.zeros(size if nd else (0,), _TENSORS[typename])
                else:
                    a = np.lib.stride_tricks.as_strided(storage[off:], shape=size,
                                                        strides=[s * storage.itemsize for s in stride]).copy()
                self.objects[idx] = a
                return a
            obj = self.objects[idx] = TorchObject(typename)
            fields = self.read()
            obj.fields = fields if isinstance(fields, dict) else {"_payload": fields}
            return obj
        raise ValueError(f"unknown torch7 object type {t} (functions and other types are not supported)")


def save(path, obj):
    with open(path, "wb") as f:
        Writer(f).write(obj)
    return path


def load(path):
    with open(path, "rb") as f:
        return Reader(f).read()
