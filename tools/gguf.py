"""Minimal GGUF v2 reader/writer (numpy only).

Format follows what the reference's parser accepts (reference models/ggml/ggml.c:19473-19493 header,
:19626-19707 KV section, :19709-19759 tensor infos + data alignment; value/tensor type enums
models/ggml/ggml.h:289-311 and :1830-1845).  This module is host-side tooling: the synthetic-model
writer (`synth.py`), the tests and the oracle use it; the device library has its own C++ parser.
"""
import struct
from collections import OrderedDict

import numpy as np

GGUF_MAGIC = 0x46554747
GGUF_VERSION = 2
DEFAULT_ALIGNMENT = 32

# ggml_type ids (reference ggml.h:289-311)
F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 0, 1, 2, 3, 6, 7, 8, 9
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K = 10, 11, 12, 13, 14, 15
TYPE_NAMES = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K"}
# (elements per block, bytes per block)
TYPE_BLOCK = {F32: (1, 4), F16: (1, 2), Q4_0: (32, 18), Q4_1: (32, 20), Q5_0: (32, 22), Q5_1: (32, 24), Q8_0: (32, 34), Q8_1: (32, 40), Q4_K: (256, 144), Q5_K: (256, 176),
              Q6_K: (256, 210), Q8_K: (256, 292)}

# gguf value types (reference ggml.h:1830-1845)
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR_FMT = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f",
               T_BOOL: "<?", T_U64: "<Q", T_I64: "<q", T_F64: "<d"}
_NP_DT = {T_U8: np.uint8, T_I8: np.int8, T_U16: np.uint16, T_I16: np.int16, T_U32: np.uint32, T_I32: np.int32,
          T_F32: np.float32, T_BOOL: np.bool_, T_U64: np.uint64, T_I64: np.int64, T_F64: np.float64}


def row_bytes(ggml_type, ne0):
    be, bb = TYPE_BLOCK[ggml_type]
    assert ne0 % be == 0, (ggml_type, ne0)
    return ne0 // be * bb


def tensor_nbytes(ggml_type, shape):
    """shape is ggml order: ne[0] = columns (K) first."""
    n = row_bytes(ggml_type, shape[0])
    for d in shape[1:]:
        n *= d
    return n


class KV:
    """Typed metadata value."""

    def __init__(self, vtype, value, elem_type=None):
        self.vtype, self.value, self.elem_type = vtype, value, elem_type


def _w_str(f, s):
    b = s if isinstance(s, bytes) else s.encode("utf-8")
    f.write(struct.pack("<Q", len(b)))
    f.write(b)


class GGUFWriter:
    """Streams a GGUF v2 file: metadata, tensor infos, then tensor data (each padded to 32)."""

    def __init__(self, path, alignment=DEFAULT_ALIGNMENT):
        self.path, self.alignment = path, alignment
        self.kv = OrderedDict()
        self.tensors = []  # (name, shape_ggml, type, nbytes, producer)

    def add(self, key, vtype, value, elem_type=None):
        self.kv[key] = KV(vtype, value, elem_type)

    def add_u32(self, k, v): self.add(k, T_U32, int(v))
    def add_f32(self, k, v): self.add(k, T_F32, float(v))
    def add_str(self, k, v): self.add(k, T_STR, v)
    def add_arr(self, k, elem_type, values): self.add(k, T_ARR, values, elem_type)

    def add_tensor(self, name, shape_ggml, ggml_type, producer):
        """producer() -> bytes-like (np.uint8 array / bytes) of exactly tensor_nbytes bytes."""
        nbytes = tensor_nbytes(ggml_type, shape_ggml)
        self.tensors.append((name, tuple(int(x) for x in shape_ggml), ggml_type, nbytes, producer))

    def _pad(self, n):
        a = self.alignment
        return (n + a - 1) // a * a

    def write(self):
        with open(self.path, "wb") as f:
            f.write(struct.pack("<IIQQ", GGUF_MAGIC, GGUF_VERSION, len(self.tensors), len(self.kv)))
            for key, kv in self.kv.items():
                _w_str(f, key)
                f.write(struct.pack("<I", kv.vtype))
                if kv.vtype == T_STR:
                    _w_str(f, kv.value)
                elif kv.vtype == T_ARR:
                    f.write(struct.pack("<IQ", kv.elem_type, len(kv.value)))
                    if kv.elem_type == T_STR:
                        for s in kv.value:
                            _w_str(f, s)
                    else:
                        f.write(np.asarray(kv.value, dtype=_NP_DT[kv.elem_type]).tobytes())
                else:
                    f.write(struct.pack(_SCALAR_FMT[kv.vtype], kv.value))
            off = 0
            for name, shape, t, nbytes, _ in self.tensors:
                _w_str(f, name)
                f.write(struct.pack("<I", len(shape)))
                f.write(struct.pack("<%dQ" % len(shape), *shape))
                f.write(struct.pack("<IQ", t, off))
                off += self._pad(nbytes)
            pos = f.tell()
            f.write(b"\0" * (self._pad(pos) - pos))
            for name, shape, t, nbytes, producer in self.tensors:
                data = producer()
                mv = memoryview(np.ascontiguousarray(data)).cast("B") if not isinstance(data, (bytes, bytearray)) else data
                assert len(mv) == nbytes, (name, len(mv), nbytes)
                f.write(mv)
                f.write(b"\0" * (self._pad(nbytes) - nbytes))


class GGUFFile:
    """Read-only mmap view: .kv (dict key -> python value), .tensors (name -> (shape_ggml, type, np.uint8 view))."""

    def __init__(self, path):
        self.path = path
        self.data = np.memmap(path, dtype=np.uint8, mode="r")
        self.pos = 0
        magic, self.version, n_tensors, n_kv = self._unpack("<IIQQ")
        if magic != GGUF_MAGIC:
            raise ValueError("not a GGUF file: %s" % path)
        if self.version == 1:
            raise ValueError("GGUF v1 not supported by this reader")
        self.kv = OrderedDict()
        for _ in range(n_kv):
            key = self._str().decode("utf-8")
            (vtype,) = self._unpack("<I")
            self.kv[key] = self._value(vtype)
        infos = []
        for _ in range(n_tensors):
            name = self._str().decode("utf-8")
            (nd,) = self._unpack("<I")
            shape = self._unpack("<%dQ" % nd)
            t, off = self._unpack("<IQ")
            infos.append((name, tuple(shape), t, off))
        align = int(self.kv.get("general.alignment", DEFAULT_ALIGNMENT))
        base = (self.pos + align - 1) // align * align
        self.tensors = OrderedDict()
        for name, shape, t, off in infos:
            n = tensor_nbytes(t, shape)
            self.tensors[name] = (shape, t, self.data[base + off: base + off + n])

    def _unpack(self, fmt):
        n = struct.calcsize(fmt)
        out = struct.unpack(fmt, self.data[self.pos:self.pos + n].tobytes())
        self.pos += n
        return out

    def _str(self):
        (n,) = self._unpack("<Q")
        s = self.data[self.pos:self.pos + n].tobytes()
        self.pos += n
        return s

    def _value(self, vtype):
        if vtype == T_STR:
            return self._str().decode("utf-8", errors="replace")
        if vtype == T_ARR:
            et, n = self._unpack("<IQ")
            if et == T_STR:
                return [self._str() for _ in range(n)]
            dt = np.dtype(_NP_DT[et])
            arr = np.frombuffer(self.data[self.pos:self.pos + n * dt.itemsize].tobytes(), dtype=dt)
            self.pos += n * dt.itemsize
            return arr
        return self._unpack(_SCALAR_FMT[vtype])[0]


class LegacyGgmlFile:
    """Reader of the pre-GGUF GGML container of the reference's legacy model loaders (here: gpt2, models/llms/gpt2.cc:61-381):
    magic 0x67676d6c, 6 x i32 hparams (n_vocab, n_ctx, n_embd, n_head, n_layer, ftype), vocab (i32 count, then u32 len +
    bytes each), tensors until EOF: i32 n_dims, i32 name_len, i32 type, dims, name, data (unaligned)."""

    def __init__(self, path, mpt=False):
        import struct
        raw = np.memmap(path, dtype=np.uint8, mode="r")
        off = 0

        def rd(fmt):
            nonlocal off
            v = struct.unpack_from("<" + fmt, raw, off)
            off += struct.calcsize("<" + fmt)
            return v

        (magic,) = rd("I")
        if magic != 0x67676d6c:
            raise ValueError("not a legacy GGML file")
        if mpt:   # models/llms/mpt.cc:70-84: d_model, max_seq_len, n_heads, n_layers, n_vocab, alibi_bias_max, clip_qkv, ftype; no count
            n_embd, n_ctx, n_head, n_layer, n_vocab, alibi_bias_max, clip_qkv, ftype = rd("5i2fi")
            self.hparams = dict(n_vocab=n_vocab, n_ctx=n_ctx, n_embd=n_embd, n_head=n_head, n_layer=n_layer, ftype=ftype % 1000,
                                alibi_bias_max=alibi_bias_max, clip_qkv=clip_qkv)
            nv = n_vocab
        else:
            n_vocab, n_ctx, n_embd, n_head, n_layer, ftype = rd("6i")
            self.hparams = dict(n_vocab=n_vocab, n_ctx=n_ctx, n_embd=n_embd, n_head=n_head, n_layer=n_layer, ftype=ftype % 1000)
            (nv,) = rd("i")
        self.vocab = []
        for _ in range(nv):
            (ln,) = rd("I")
            self.vocab.append(bytes(raw[off:off + ln]))
            off += ln
        self.tensors = {}
        while off < raw.size:
            n_dims, name_len, ttype = rd("3i")
            dims = rd("%di" % n_dims)
            name = bytes(raw[off:off + name_len]).decode("ascii")
            off += name_len
            n_el = int(np.prod(dims))
            be, bb = TYPE_BLOCK[ttype]
            nbytes = n_el // be * bb
            self.tensors[name] = (tuple(dims), ttype, raw[off:off + nbytes])
            off += nbytes
