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
    from tools import gguf as G
    x = np.ascontiguousarray(x, dtype=np.float32)
    vt = traits(weight_type).vec_dot_type
    be, bb = G.TYPE_BLOCK[vt]
    out = np.zeros(x.size // be * bb, dtype=np.uint8)
    _FROM_FLOAT(traits(vt).from_float)(_fptr(x), out.ctypes.data_as(c_void_p), x.size)
    return out, vt


def quantize_chunk(x, ggml_type):
    """f32 [..., K] -> uint8 [..., row_bytes] through the reference's OWN quantizer, `ggml_quantize_chunk` (reference
    models/ggml/ggml.c:19319; K-quants: models/ggml/k_quants.c:600-1115 — make_qkx1_quants / make_qx_quants scale searches,
    Q6_K with its `iscale = -128.f/max_scale`, which gives negative block scales and negative d).  This is what the files the
    reference reads in the field were produced by (llama.cpp's quantize tool calls it per tensor chunk)."""
    from tools import gguf as G
    x = np.ascontiguousarray(x, dtype=np.float32)
    lead, K = x.shape[:-1], x.shape[-1]
    be, bb = G.TYPE_BLOCK[int(ggml_type)]
    assert K % be == 0
    n = x.size
    out = np.zeros(n // be * bb, dtype=np.uint8)
    hist = (ctypes.c_int64 * 16)()
    L = lib()
    L.ggml_quantize_chunk.argtypes = [c_int, POINTER(c_float), c_void_p, c_int, c_int, POINTER(ctypes.c_int64)]
    L.ggml_quantize_chunk.restype = c_size_t
    step = max(be, (1 << 24) // be * be)     # int-sized chunks, as the quantize tool feeds it
    flat = x.reshape(-1)
    for start in range(0, n, step):
        cnt = min(step, n - start)
        got = L.ggml_quantize_chunk(int(ggml_type), _fptr(flat), out.ctypes.data_as(c_void_p), start, cnt, hist)
        assert got == cnt // be * bb, (got, cnt)
    return out.reshape(lead + (K // be * bb,))


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


def open_llm(path, model_type=None, **cfg):
    """Whole-model oracle: the reference CPU implementation driven through the same Python host mirror.
    model_type is needed for legacy (pre-GGUF) files, e.g. "gpt2"."""
    from ctransformers_amd.llm import LLM, Config
    lib()  # existence check
    return LLM(path, model_type, config=Config(**cfg), lib=REF_LIB)


# ---------------------------------------------------------------------------------------------------------------------
# op-level oracle through the reference's own graph executor (ggml.h public API, exported by the .so)
# ---------------------------------------------------------------------------------------------------------------------
class GgmlOps:
    """Runs single ggml ops of the reference build on numpy data: rms_norm, rope, soft_max, scale, f16 mat-mul."""

    def __init__(self, mem_mb=256):
        L = lib()
        self.L = L
        vp = c_void_p
        L.ggml_new_tensor_1d.argtypes = [vp, c_int, ctypes.c_int64]
        L.ggml_new_tensor_2d.argtypes = [vp, c_int, ctypes.c_int64, ctypes.c_int64]
        L.ggml_new_tensor_3d.argtypes = [vp, c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        for f in (L.ggml_new_tensor_1d, L.ggml_new_tensor_2d, L.ggml_new_tensor_3d, L.ggml_get_data, L.ggml_new_graph,
                  L.ggml_rms_norm, L.ggml_soft_max, L.ggml_scale, L.ggml_mul_mat, L.ggml_rope_custom_inplace,
                  L.ggml_new_f32, L.ggml_mul, L.ggml_add, L.ggml_silu, L.ggml_cpy, L.ggml_diag_mask_inf, L.ggml_gelu,
                  L.ggml_norm):
            f.restype = vp
        L.ggml_get_data.argtypes = [vp]
        L.ggml_new_graph.argtypes = [vp]
        L.ggml_rms_norm.argtypes = [vp, vp, c_float]
        L.ggml_soft_max.argtypes = [vp, vp]
        L.ggml_scale.argtypes = [vp, vp, vp]
        L.ggml_mul_mat.argtypes = [vp, vp, vp]
        L.ggml_mul.argtypes = [vp, vp, vp]
        L.ggml_add.argtypes = [vp, vp, vp]
        L.ggml_cpy.argtypes = [vp, vp, vp]
        L.ggml_silu.argtypes = [vp, vp]
        L.ggml_gelu.argtypes = [vp, vp]
        L.ggml_norm.argtypes = [vp, vp, c_float]
        L.ggml_new_f32.argtypes = [vp, c_float]
        L.ggml_rope_custom_inplace.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_float, c_float]
        L.ggml_build_forward_expand.argtypes = [vp, vp]
        L.ggml_graph_compute_with_ctx.argtypes = [vp, vp, c_int]
        L.ggml_free.argtypes = [vp]
        self.mem = mem_mb << 20

    def _ctx(self):
        return self.L.ggml_init(_InitParams(self.mem, None, False))

    def _tensor(self, ctx, arr, ggml_type=0):
        arr = np.ascontiguousarray(arr)
        shape = arr.shape[::-1]  # ggml order: ne[0] fastest
        new = [self.L.ggml_new_tensor_1d, self.L.ggml_new_tensor_2d, self.L.ggml_new_tensor_3d][len(shape) - 1]
        t = new(ctx, ggml_type, *[int(s) for s in shape])
        ctypes.memmove(self.L.ggml_get_data(t), arr.ctypes.data, arr.nbytes)
        return t

    def _run(self, ctx, t, shape, dtype=np.float32, threads=1):
        gf = self.L.ggml_new_graph(ctx)
        self.L.ggml_build_forward_expand(gf, t)
        self.L.ggml_graph_compute_with_ctx(ctx, gf, threads)
        n = int(np.prod(shape))
        out = np.empty(n, dtype=dtype)
        ctypes.memmove(out.ctypes.data, self.L.ggml_get_data(t), out.nbytes)
        self.L.ggml_free(ctx)
        return out.reshape(shape)

    def rms_norm_mul(self, x, w, eps):
        ctx = self._ctx()
        t = self.L.ggml_mul(ctx, self.L.ggml_rms_norm(ctx, self._tensor(ctx, x.astype(np.float32)), eps),
                            self._tensor(ctx, w.astype(np.float32)))
        return self._run(ctx, t, x.shape)

    def rope(self, x, n_past, mode=0, freq_base=10000.0, freq_scale=1.0):
        """x: [N, n_head, head_dim] f32 (numpy order) -> rotated copy."""
        ctx = self._ctx()
        t = self._tensor(ctx, x.astype(np.float32))
        r = self.L.ggml_rope_custom_inplace(ctx, t, int(n_past), int(x.shape[-1]), mode, 0, freq_base, freq_scale)
        return self._run(ctx, r, x.shape)

    def scale_softmax(self, scores, scale):
        """scores: [rows, n] f32 -> soft_max(scale * scores) rows."""
        ctx = self._ctx()
        t = self._tensor(ctx, scores.astype(np.float32))
        s = self.L.ggml_scale(ctx, t, self.L.ggml_new_f32(ctx, float(scale)))
        return self._run(ctx, self.L.ggml_soft_max(ctx, s), scores.shape)

    def mul_mat_f16(self, a_f16, b_f32):
        """a: [M, K] fp16 (uint16 bits / np.float16), b: [N, K] f32 -> [N, M] f32 = ggml_mul_mat(a, b)."""
        ctx = self._ctx()
        ta = self._tensor(ctx, np.ascontiguousarray(a_f16).view(np.uint16), 1)
        tb = self._tensor(ctx, b_f32.astype(np.float32))
        return self._run(ctx, self.L.ggml_mul_mat(ctx, ta, tb), (b_f32.shape[0], a_f16.shape[0]))

    def gelu(self, x):
        ctx = self._ctx()
        return self._run(ctx, self.L.ggml_gelu(ctx, self._tensor(ctx, x.astype(np.float32))), x.shape)

    def norm_mul_add(self, x, w, b, eps):
        """LayerNorm as the falcon graph builds it: ggml_add(ggml_mul(ggml_norm(x), w), b)  (llama.cpp:2611-2614)."""
        ctx = self._ctx()
        t = self.L.ggml_add(ctx, self.L.ggml_mul(ctx, self.L.ggml_norm(ctx, self._tensor(ctx, x.astype(np.float32)), eps),
                                                  self._tensor(ctx, w.astype(np.float32))),
                            self._tensor(ctx, b.astype(np.float32)))
        return self._run(ctx, t, x.shape)

    def silu_mul(self, g, u):
        ctx = self._ctx()
        t = self.L.ggml_mul(ctx, self.L.ggml_silu(ctx, self._tensor(ctx, g.astype(np.float32))),
                            self._tensor(ctx, u.astype(np.float32)))
        return self._run(ctx, t, g.shape)
