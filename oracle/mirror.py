"""TEST INFRASTRUCTURE ONLY — Python driver of the C restatement (oracle/mirror.c, built by oracle/Makefile into
oracle/_build/libmirror.so).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_void_p

import numpy as np

from tools import gguf as G

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmirror.so")
_lib = None


def available():
    return os.path.isfile(LIB)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB)
        _lib.mir_f16_to_f32.argtypes = [ctypes.c_uint16]
        _lib.mir_f16_to_f32.restype = c_float
        _lib.mir_f32_to_f16.argtypes = [c_float]
        _lib.mir_f32_to_f16.restype = ctypes.c_uint16
        for n in ("q4_K_q8_K", "q5_K_q8_K", "q6_K_q8_K", "q8_0_q8_0", "q4_0_q8_0"):
            f = getattr(_lib, "mir_vec_dot_" + n)
            f.argtypes = [c_int, c_void_p, c_void_p]
            f.restype = c_float
        _lib.mir_matvec.argtypes = [c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]
        _lib.mir_dequantize_row.argtypes = [c_int, c_void_p, c_void_p, c_int]
        _lib.mir_quantize_row_q8_K.argtypes = [c_void_p, c_void_p, c_int]
        _lib.mir_quantize_row_q8_0.argtypes = [c_void_p, c_void_p, c_int]
        _lib.mir_llama_eval.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
        _lib.mir_llama_eval.restype = c_int
        _lib.mir_llama_eval_range.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p]
        _lib.mir_llama_eval_range.restype = c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


def quantize_q8_K(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(x.size // 256 * 292, dtype=np.uint8)
    lib().mir_quantize_row_q8_K(_p(x), _p(out), x.size)
    return out


def quantize_q8_0(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(x.size // 32 * 34, dtype=np.uint8)
    lib().mir_quantize_row_q8_0(_p(x), _p(out), x.size)
    return out


def dequantize(raw, ggml_type, K):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    rows = raw.reshape(-1, raw.shape[-1])
    out = np.zeros((rows.shape[0], K), dtype=np.float32)
    for r in range(rows.shape[0]):
        row = np.ascontiguousarray(rows[r])
        lib().mir_dequantize_row(int(ggml_type), _p(row), _p(out[r]), K)
    return out


def matvec(ggml_type, wraw, x, K):
    w = np.ascontiguousarray(wraw, dtype=np.uint8)
    x = np.ascontiguousarray(x, dtype=np.float32)
    M = w.shape[0]
    out = np.zeros(M, dtype=np.float32)
    lib().mir_matvec(int(ggml_type), _p(w), M, K, _p(x), _p(out))
    return out


class _Model(Structure):
    _fields_ = [("n_vocab", c_int), ("n_embd", c_int), ("n_head", c_int), ("n_head_kv", c_int), ("n_layer", c_int),
                ("n_ff", c_int), ("n_ctx", c_int), ("rms_eps", c_float), ("rope_freq_base", c_float),
                ("rope_freq_scale", c_float),
                ("tok_embd", c_void_p), ("tok_embd_type", c_int), ("output_norm", c_void_p), ("output", c_void_p),
                ("output_type", c_int), ("attn_norm", c_void_p), ("ffn_norm", c_void_p), ("wq", c_void_p),
                ("wk", c_void_p), ("wv", c_void_p), ("wo", c_void_p), ("w_gate", c_void_p), ("w_up", c_void_p),
                ("w_down", c_void_p), ("t_wq", c_void_p), ("t_wk", c_void_p), ("t_wv", c_void_p), ("t_wo", c_void_p),
                ("t_gate", c_void_p), ("t_up", c_void_p), ("t_down", c_void_p), ("kcache", c_void_p),
                ("vcache", c_void_p)]


class MirrorLlama:
    """The C restatement driven over a GGUF file: same eval(tokens, n_past) contract as the ABI's batch_eval."""

    def __init__(self, path, context_length=512):
        self.f = G.GGUFFile(path)
        kv = self.f.kv
        a = "llama."
        self.n_embd = int(kv[a + "embedding_length"])
        self.n_head = int(kv[a + "attention.head_count"])
        self.n_head_kv = int(kv.get(a + "attention.head_count_kv", self.n_head))
        self.n_layer = int(kv[a + "block_count"])
        self.n_ff = int(kv[a + "feed_forward_length"])
        self.n_vocab = len(kv["tokenizer.ggml.tokens"])
        self.n_ctx = context_length
        self._keep = []
        m = _Model()
        m.n_vocab, m.n_embd, m.n_head, m.n_head_kv = self.n_vocab, self.n_embd, self.n_head, self.n_head_kv
        m.n_layer, m.n_ff, m.n_ctx = self.n_layer, self.n_ff, self.n_ctx
        m.rms_eps = float(kv[a + "attention.layer_norm_rms_epsilon"])
        m.rope_freq_base = float(kv.get(a + "rope.freq_base", 10000.0))
        m.rope_freq_scale = 1.0

        def tensor(name):
            shape, t, data = self.f.tensors[name]
            arr = np.ascontiguousarray(data)
            self._keep.append(arr)
            return arr.ctypes.data, t

        m.tok_embd, m.tok_embd_type = tensor("token_embd.weight")
        m.output_norm, _ = tensor("output_norm.weight")
        m.output, m.output_type = tensor("output.weight")

        def per_layer(fmt, types_field=None):
            ptrs = (c_void_p * self.n_layer)()
            types = (c_int * self.n_layer)()
            for i in range(self.n_layer):
                ptrs[i], types[i] = tensor(fmt % i)
            self._keep += [ptrs, types]
            return ctypes.cast(ptrs, c_void_p), ctypes.cast(types, c_void_p)

        m.attn_norm, _ = per_layer("blk.%d.attn_norm.weight")
        m.ffn_norm, _ = per_layer("blk.%d.ffn_norm.weight")
        m.wq, m.t_wq = per_layer("blk.%d.attn_q.weight")
        m.wk, m.t_wk = per_layer("blk.%d.attn_k.weight")
        m.wv, m.t_wv = per_layer("blk.%d.attn_v.weight")
        m.wo, m.t_wo = per_layer("blk.%d.attn_output.weight")
        m.w_gate, m.t_gate = per_layer("blk.%d.ffn_gate.weight")
        m.w_up, m.t_up = per_layer("blk.%d.ffn_up.weight")
        m.w_down, m.t_down = per_layer("blk.%d.ffn_down.weight")
        g = self.n_embd // self.n_head * self.n_head_kv
        self.kc = np.zeros(self.n_layer * self.n_ctx * g, dtype=np.uint16)
        self.vc = np.zeros(self.n_layer * self.n_ctx * g, dtype=np.uint16)
        m.kcache, m.vcache = self.kc.ctypes.data, self.vc.ctypes.data
        self.m = m
        self.logits = np.zeros(self.n_vocab, dtype=np.float32)
        self.embeddings = np.zeros(self.n_embd, dtype=np.float32)

    def eval(self, tokens, n_past):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        lib().mir_llama_eval(ctypes.byref(self.m), _p(t), len(t), int(n_past), _p(self.logits), _p(self.embeddings))
        return self.logits


# op-level entry points of the restatement (same semantics as oracle/ref.py GgmlOps, for cross-checks)

    def eval_range(self, tokens, n_past, l0, l1, x_in=None):
        """Layers [l0, l1) only (pipeline stage oracle): returns x_out [n, n_embd] for an inner stage, logits for the last."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        xi = None if x_in is None else np.ascontiguousarray(x_in, dtype=np.float32)
        xo = np.zeros((len(t), self.n_embd), dtype=np.float32)
        lib().mir_llama_eval_range(ctypes.byref(self.m), _p(t), len(t), int(n_past), int(l0), int(l1),
                                   None if xi is None else _p(xi), _p(xo), _p(self.logits), _p(self.embeddings))
        return self.logits if l1 == self.n_layer else xo


def rms_norm_mul(x, w, eps):
    x = np.ascontiguousarray(x, dtype=np.float32); w = np.ascontiguousarray(w, dtype=np.float32)
    y = np.zeros_like(x)
    L = lib()
    L.mir_rms_norm_mul.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_float]
    L.mir_rms_norm_mul(_p(x), _p(w), _p(y), x.size, float(eps))
    return y


def rope(x, pos, freq_base=10000.0, freq_scale=1.0):
    """x: [n_head, head_dim]"""
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    L = lib()
    L.mir_rope.argtypes = [c_void_p, c_int, c_int, c_int, c_float, c_float]
    L.mir_rope(_p(y), y.shape[0], y.shape[1], int(pos), float(freq_base), float(freq_scale))
    return y


def norm_mul_add(x, w, b, eps):
    x, w, b = (np.ascontiguousarray(v, dtype=np.float32) for v in (x, w, b))
    y = np.zeros_like(x)
    lib().mir_norm_mul_add.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float]
    lib().mir_norm_mul_add(_p(x), _p(w), _p(b), _p(y), x.size, float(eps))
    return y


def rope_neox(x, pos, freq_base=10000.0, freq_scale=1.0):
    """x: [n_head, head_dim] f32 -> rotated copy (neox mode)."""
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().mir_rope_neox.argtypes = [c_void_p, c_int, c_int, c_int, c_float, c_float]
    lib().mir_rope_neox(_p(y), y.shape[0], y.shape[1], int(pos), float(freq_base), float(freq_scale))
    return y


def gelu(x):
    lib().mir_gelu.restype = c_float
    lib().mir_gelu.argtypes = [c_float]
    return np.array([lib().mir_gelu(float(v)) for v in np.asarray(x, dtype=np.float32).ravel()], dtype=np.float32).reshape(np.shape(x))


class _Falcon(Structure):
    _fields_ = [("n_vocab", c_int), ("n_embd", c_int), ("n_head", c_int), ("n_head_kv", c_int), ("n_layer", c_int),
                ("n_ff", c_int), ("n_ctx", c_int), ("norm_eps", c_float), ("rope_freq_base", c_float),
                ("rope_freq_scale", c_float), ("tok_embd", c_void_p), ("tok_embd_type", c_int),
                ("output_norm", c_void_p), ("output_norm_b", c_void_p), ("output", c_void_p), ("output_type", c_int),
                ("attn_norm", c_void_p), ("attn_norm_b", c_void_p), ("attn_norm_2", c_void_p), ("attn_norm_2_b", c_void_p),
                ("wqkv", c_void_p), ("wo", c_void_p), ("w_up", c_void_p), ("w_down", c_void_p),
                ("t_wqkv", c_void_p), ("t_wo", c_void_p), ("t_up", c_void_p), ("t_down", c_void_p),
                ("kcache", c_void_p), ("vcache", c_void_p)]


class MirrorFalcon:
    """The C restatement of llm_build_falcon driven over a GGUF file (same eval contract as MirrorLlama)."""

    def __init__(self, path, context_length=512):
        self.f = G.GGUFFile(path)
        kv = self.f.kv
        a = "falcon."
        self.n_embd = int(kv[a + "embedding_length"])
        self.n_head = int(kv[a + "attention.head_count"])
        self.n_head_kv = int(kv.get(a + "attention.head_count_kv", self.n_head))
        self.n_layer = int(kv[a + "block_count"])
        self.n_ff = int(kv[a + "feed_forward_length"])
        self.n_vocab = len(kv["tokenizer.ggml.tokens"])
        self.n_ctx = context_length
        self._keep = []
        m = _Falcon()
        m.n_vocab, m.n_embd, m.n_head, m.n_head_kv = self.n_vocab, self.n_embd, self.n_head, self.n_head_kv
        m.n_layer, m.n_ff, m.n_ctx = self.n_layer, self.n_ff, self.n_ctx
        m.norm_eps = float(kv[a + "attention.layer_norm_epsilon"])
        m.rope_freq_base = float(kv.get(a + "rope.freq_base", 10000.0))
        m.rope_freq_scale = 1.0

        def tensor(name, optional=False):
            if optional and name not in self.f.tensors:
                return None, 0
            shape, t, data = self.f.tensors[name]
            arr = np.ascontiguousarray(data)
            self._keep.append(arr)
            return arr.ctypes.data, t

        m.tok_embd, m.tok_embd_type = tensor("token_embd.weight")
        m.output_norm, _ = tensor("output_norm.weight")
        m.output_norm_b, _ = tensor("output_norm.bias")
        m.output, m.output_type = tensor("output.weight")

        def per_layer(fmt, optional=False):
            ptrs = (c_void_p * self.n_layer)()
            types = (c_int * self.n_layer)()
            for i in range(self.n_layer):
                ptrs[i], types[i] = tensor(fmt % i, optional)
            self._keep += [ptrs, types]
            return ctypes.cast(ptrs, c_void_p), ctypes.cast(types, c_void_p)

        m.attn_norm, _ = per_layer("blk.%d.attn_norm.weight")
        m.attn_norm_b, _ = per_layer("blk.%d.attn_norm.bias")
        m.attn_norm_2, _ = per_layer("blk.%d.attn_norm_2.weight", True)
        m.attn_norm_2_b, _ = per_layer("blk.%d.attn_norm_2.bias", True)
        m.wqkv, m.t_wqkv = per_layer("blk.%d.attn_qkv.weight")
        m.wo, m.t_wo = per_layer("blk.%d.attn_output.weight")
        m.w_up, m.t_up = per_layer("blk.%d.ffn_up.weight")
        m.w_down, m.t_down = per_layer("blk.%d.ffn_down.weight")
        g = self.n_embd // self.n_head * self.n_head_kv
        self.kc = np.zeros(self.n_layer * self.n_ctx * g, dtype=np.uint16)
        self.vc = np.zeros(self.n_layer * self.n_ctx * g, dtype=np.uint16)
        m.kcache, m.vcache = self.kc.ctypes.data, self.vc.ctypes.data
        self.m = m
        self.logits = np.zeros(self.n_vocab, dtype=np.float32)
        self.embeddings = np.zeros(self.n_embd, dtype=np.float32)
        lib().mir_falcon_eval.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
        lib().mir_falcon_eval.restype = c_int

    def eval(self, tokens, n_past):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        lib().mir_falcon_eval(ctypes.byref(self.m), _p(t), len(t), int(n_past), _p(self.logits), _p(self.embeddings))
        return self.logits


class _Gpt2(Structure):
    _fields_ = [("n_vocab", c_int), ("n_ctx", c_int), ("n_embd", c_int), ("n_head", c_int), ("n_layer", c_int),
                ("wte", c_void_p), ("wte_type", c_int), ("wpe", c_void_p), ("lm_head", c_void_p), ("lm_head_type", c_int),
                ("ln_f_g", c_void_p), ("ln_f_b", c_void_p),
                ("ln_1_g", c_void_p), ("ln_1_b", c_void_p), ("ln_2_g", c_void_p), ("ln_2_b", c_void_p),
                ("c_attn_w", c_void_p), ("c_attn_b", c_void_p), ("c_proj_w", c_void_p), ("c_proj_b", c_void_p),
                ("fc_w", c_void_p), ("fc_b", c_void_p), ("proj_w", c_void_p), ("proj_b", c_void_p),
                ("wtype", c_int), ("memory_k", c_void_p), ("memory_v", c_void_p)]


class MirrorGpt2:
    """The C restatement of gpt2_eval over a legacy GGML file (n_ctx comes from the file, like the reference)."""

    def __init__(self, path):
        self.f = G.LegacyGgmlFile(path)
        hp = self.f.hparams
        self.n_vocab, self.n_ctx, self.n_embd, self.n_layer = hp["n_vocab"], hp["n_ctx"], hp["n_embd"], hp["n_layer"]
        self._keep = []
        m = _Gpt2()
        m.n_vocab, m.n_ctx, m.n_embd, m.n_head, m.n_layer = hp["n_vocab"], hp["n_ctx"], hp["n_embd"], hp["n_head"], hp["n_layer"]

        def tensor(name):
            shape, t, data = self.f.tensors[name]
            arr = np.ascontiguousarray(data)
            self._keep.append(arr)
            return arr.ctypes.data, t

        m.wte, m.wte_type = tensor("model/wte")
        m.wpe, _ = tensor("model/wpe")
        if "model/lm_head" in self.f.tensors:
            m.lm_head, m.lm_head_type = tensor("model/lm_head")
        else:
            m.lm_head, m.lm_head_type = m.wte, m.wte_type
        m.ln_f_g, _ = tensor("model/ln_f/g")
        m.ln_f_b, _ = tensor("model/ln_f/b")

        def per_layer(fmt):
            ptrs = (c_void_p * self.n_layer)()
            t = 0
            for i in range(self.n_layer):
                ptrs[i], t = tensor(fmt % i)
            self._keep.append(ptrs)
            return ctypes.cast(ptrs, c_void_p), t

        m.ln_1_g, _ = per_layer("model/h%d/ln_1/g")
        m.ln_1_b, _ = per_layer("model/h%d/ln_1/b")
        m.ln_2_g, _ = per_layer("model/h%d/ln_2/g")
        m.ln_2_b, _ = per_layer("model/h%d/ln_2/b")
        m.c_attn_w, m.wtype = per_layer("model/h%d/attn/c_attn/w")
        m.c_attn_b, _ = per_layer("model/h%d/attn/c_attn/b")
        m.c_proj_w, _ = per_layer("model/h%d/attn/c_proj/w")
        m.c_proj_b, _ = per_layer("model/h%d/attn/c_proj/b")
        m.fc_w, _ = per_layer("model/h%d/mlp/c_fc/w")
        m.fc_b, _ = per_layer("model/h%d/mlp/c_fc/b")
        m.proj_w, _ = per_layer("model/h%d/mlp/c_proj/w")
        m.proj_b, _ = per_layer("model/h%d/mlp/c_proj/b")
        self.mk = np.zeros(self.n_layer * self.n_ctx * self.n_embd, dtype=np.float32)
        self.mv = np.zeros(self.n_layer * self.n_ctx * self.n_embd, dtype=np.float32)
        m.memory_k, m.memory_v = self.mk.ctypes.data, self.mv.ctypes.data
        self.m = m
        self.logits = np.zeros(self.n_vocab, dtype=np.float32)
        lib().mir_gpt2_eval.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
        lib().mir_gpt2_eval.restype = c_int

    def eval(self, tokens, n_past):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        lib().mir_gpt2_eval(ctypes.byref(self.m), _p(t), len(t), int(n_past), _p(self.logits))
        return self.logits


class _Mpt(Structure):
    _fields_ = [("n_vocab", c_int), ("n_ctx", c_int), ("n_embd", c_int), ("n_head", c_int), ("n_layer", c_int),
                ("alibi_bias_max", ctypes.c_float), ("clip_qkv", ctypes.c_float), ("wte", c_void_p), ("wtype", c_int),
                ("norm_f", c_void_p), ("norm_1", c_void_p), ("norm_2", c_void_p),
                ("wqkv", c_void_p), ("out_proj", c_void_p), ("up_proj", c_void_p), ("down_proj", c_void_p),
                ("memory_k", c_void_p), ("memory_v", c_void_p)]


class MirrorMpt:
    """The C restatement of mpt_eval over a legacy GGML file with the MPT header; n_ctx = min(max_seq_len, n_ctx or 2048)."""

    def __init__(self, path, n_ctx=0):
        self.f = G.LegacyGgmlFile(path, mpt=True)
        hp = self.f.hparams
        self.n_vocab, self.n_embd, self.n_layer = hp["n_vocab"], hp["n_embd"], hp["n_layer"]
        self.n_ctx = min(hp["n_ctx"], n_ctx if n_ctx > 0 else 2048)
        self._keep = []
        m = _Mpt()
        m.n_vocab, m.n_ctx, m.n_embd, m.n_head, m.n_layer = hp["n_vocab"], self.n_ctx, hp["n_embd"], hp["n_head"], hp["n_layer"]
        m.alibi_bias_max, m.clip_qkv = hp["alibi_bias_max"], hp["clip_qkv"]

        def tensor(name):
            shape, t, data = self.f.tensors[name]
            arr = np.ascontiguousarray(data)
            self._keep.append(arr)
            return arr.ctypes.data, t

        m.wte, m.wtype = tensor("transformer.wte.weight")
        m.norm_f, _ = tensor("transformer.norm_f.weight")

        def per_layer(fmt):
            ptrs = (c_void_p * self.n_layer)()
            for i in range(self.n_layer):
                ptrs[i], _ = tensor(fmt % i)
            self._keep.append(ptrs)
            return ctypes.cast(ptrs, c_void_p)

        m.norm_1 = per_layer("transformer.blocks.%d.norm_1.weight")
        m.norm_2 = per_layer("transformer.blocks.%d.norm_2.weight")
        m.wqkv = per_layer("transformer.blocks.%d.attn.Wqkv.weight")
        m.out_proj = per_layer("transformer.blocks.%d.attn.out_proj.weight")
        m.up_proj = per_layer("transformer.blocks.%d.ffn.up_proj.weight")
        m.down_proj = per_layer("transformer.blocks.%d.ffn.down_proj.weight")
        self.mk = np.zeros(self.n_layer * self.n_ctx * self.n_embd, dtype=np.uint16)
        self.mv = np.zeros(self.n_layer * self.n_ctx * self.n_embd, dtype=np.uint16)
        m.memory_k, m.memory_v = self.mk.ctypes.data, self.mv.ctypes.data
        self.m = m
        self.logits = np.zeros(self.n_vocab, dtype=np.float32)
        lib().mir_mpt_eval.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
        lib().mir_mpt_eval.restype = c_int

    def eval(self, tokens, n_past):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        lib().mir_mpt_eval(ctypes.byref(self.m), _p(t), len(t), int(n_past), _p(self.logits))
        return self.logits
