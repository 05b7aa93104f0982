"""TEST INFRASTRUCTURE ONLY — ctypes access to the REAL reference build (oracle/_ref/libctransformers_ref.so,
built by oracle/Makefile from /root/reference sources, never copied into this repo).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.  It gives
  * the whole-model oracle: load the .so through ctransformers_amd.llm.LLM(lib=REF_LIB) (same 17-symbol ABI);
  * op-level oracles through the ggml internals the .so exports (reference models/ggml/ggml.h:1984-1996
    `ggml_internal_get_type_traits`; k_quants.h:130-165 quantize_row_* / dequantize_row_* / ggml_vec_dot_*).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_bool, c_char_p, c_float, c_int, c_size_t, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(HERE, "_ref", "libctransformers_ref.so")

_lib = None


def available():
    return os.path.isfile(REF_LIB)


class _InitParams(Structure):
    _fields_ = [("mem_size", c_size_t), ("mem_buffer", c_void_p), ("no_alloc", c_bool)]


class _TypeTraits(Structure):
    _fields_ = [("type_name", c_char_p), ("blck_size", c_int), ("type_size", c_size_t), ("is_quantized", c_bool),
                ("to_float", c_void_p), ("from_float", c_void_p), ("from_float_reference", c_void_p),
                ("vec_dot", c_void_p), ("vec_dot_type", c_int)]


_TO_FLOAT = ctypes.CFUNCTYPE(None, c_void_p, POINTER(c_float), c_int)
_FROM_FLOAT = ctypes.CFUNCTYPE(None, POINTER(c_float), c_void_p, c_int)
_VEC_DOT = ctypes.CFUNCTYPE(None, c_int, POINTER(c_float), c_void_p, c_void_p)


def lib():
    """The reference .so with ggml_init() called once (fills the fp16 tables, reference ggml.c:4310-4340)."""
    global _lib
    if _lib is None:
        if not available():
            raise OSError("reference oracle not built: run `make -C oracle` where /root/reference exists")
        _lib = ctypes.CDLL(REF_LIB)
        _lib.ggml_init.argtypes = [_InitParams]
        _lib.ggml_init.restype = c_void_p
        _lib.ggml_init(_InitParams(0, None, False))
        _lib.ggml_internal_get_type_traits.argtypes = [c_int]
        _lib.ggml_internal_get_type_traits.restype = _TypeTraits
        _lib.ggml_fp32_to_fp16.argtypes = [c_float]
        _lib.ggml_fp32_to_fp16.restype = ctypes.c_uint16
        _lib.ggml_fp16_to_fp32.argtypes = [ctypes.c_uint16]
        _lib.ggml_fp16_to_fp32.restype = c_float
    return _lib


def traits(ggml_type):
    return lib().ggml_internal_get_type_traits(int(ggml_type))


def _fptr(a):
    return a.ctypes.data_as(POINTER(c_float))


def dequantize(raw, ggml_type, K):
    """raw uint8 [rows, row_bytes] -> f32 [rows, K] via the reference's to_float."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    rows = raw.reshape(-1, raw.shape[-1]) if raw.ndim > 1 else raw.reshape(1, -1)
    out = np.zeros((rows.shape[0], K), dtype=np.float32)
    fn = _TO_FLOAT(traits(ggml_type).to_float)
    for r in range(rows.shape[0]):
        row = np.ascontiguousarray(rows[r])
        fn(row.ctypes.data_as(c_void_p), _fptr(out[r]), K)
    return out


def quantize_activation(x, weight_type):
    """f32 [K] -> raw bytes of the weight type's vec_dot_type (Q8_K for K-quants, Q8_0 for Q4_0/Q8_0), exactly as
    ggml_compute_forward_mul_mat's INIT phase does (reference ggml.c:11141-11154)."""
    from ctransformers_amd import gguf as G
    x = np.ascontiguousarray(x, dtype=np.float32)
    vt = traits(weight_type).vec_dot_type
    be, bb = G.TYPE_BLOCK[vt]
    out = np.zeros(x.size // be * bb, dtype=np.uint8)
    _FROM_FLOAT(traits(vt).from_float)(_fptr(x), out.ctypes.data_as(c_void_p), x.size)
    return out, vt


def vec_dot(weight_type, wrow, act_raw, K):
    """One reference dot product: quantized weight row (raw bytes) x quantized activation (raw bytes)."""
    w = np.ascontiguousarray(wrow, dtype=np.uint8)
    a = np.ascontiguousarray(act_raw, dtype=np.uint8)
    s = c_float(0)
    _VEC_DOT(traits(weight_type).vec_dot)(K, ctypes.byref(s), w.ctypes.data_as(c_void_p), a.ctypes.data_as(c_void_p))
    return s.value


def matvec(weight_type, wraw, x, K):
    """rows of raw quantized weights [M, row_bytes] times f32 x [K], reference semantics (quantize x, then vec_dot)."""
    a, _ = quantize_activation(x, weight_type)
    M = wraw.shape[0]
    return np.array([vec_dot(weight_type, wraw[r], a, K) for r in range(M)], dtype=np.float32)


def open_llm(path, **cfg):
    """Whole-model oracle: the reference CPU implementation driven through the same Python host mirror."""
    from ctransformers_amd.llm import LLM, Config
    lib()  # existence check
    return LLM(path, config=Config(**cfg), lib=REF_LIB)
