"""TEST / BENCH INFRASTRUCTURE (not part of the product package): synthetic model files at real architecture shapes (no weights
are downloadable here).

* numpy block quantizers producing *valid* GGML blocks (layouts: reference models/ggml/ggml.c:888-925 for
  Q4_0/Q8_0, models/ggml/k_quants.h:76-126 for Q4_K/Q5_K/Q6_K).  They are NOT the reference's quantizers
  (k_quants.c:600-760 search for optimal scales); a model file is an *input* that both the reference and this
  framework read, so any valid block stream is a fair test vector.  `dequantize()` restates the reference's
  dequantize_row_* (k_quants.c:784-821, :984-1035, :1123-1170; ggml.c:1503-1560) and is checked against the
  reference build in tests.
* `write_llama_gguf()` / `write_falcon_gguf()`: GGUF v2 files the reference loader accepts (key list:
  reference models/ggml/llama.cpp:1562-1680; tensor names :294-328; per-tensor type mix of the *_K_M ftypes
  :4785-4850).
"""
import numpy as np

from . import gguf as G

QK_K = 256

# ----------------------------------------------------------------------------------------------------------------------
# quantizers (vectorised over blocks).  x: float32 [..., K]  ->  uint8 [..., K/block * block_bytes]
# ----------------------------------------------------------------------------------------------------------------------


def _f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16)


def _pack_scales_k4(sc, mn):
    """sc, mn: uint8 [nb, 8] (6-bit) -> uint8 [nb, 12]; inverse of get_scale_min_k4 (reference k_quants.c:306-314)."""
    q = np.zeros(sc.shape[:-1] + (12,), dtype=np.uint8)
    q[..., 0:4] = (sc[..., 0:4] & 63) | ((sc[..., 4:8] >> 4) << 6)
    q[..., 4:8] = (mn[..., 0:4] & 63) | ((mn[..., 4:8] >> 4) << 6)
    q[..., 8:12] = (sc[..., 4:8] & 0xF) | ((mn[..., 4:8] & 0xF) << 4)
    return q


def _unpack_scales_k4(q):
    sc = np.zeros(q.shape[:-1] + (8,), dtype=np.uint8)
    mn = np.zeros_like(sc)
    sc[..., 0:4] = q[..., 0:4] & 63
    mn[..., 0:4] = q[..., 4:8] & 63
    sc[..., 4:8] = (q[..., 8:12] & 0xF) | ((q[..., 0:4] >> 6) << 4)
    mn[..., 4:8] = (q[..., 8:12] >> 4) | ((q[..., 4:8] >> 6) << 4)
    return sc, mn


def _kquant_affine(x, levels):
    """Shared Q4_K/Q5_K scale search: x [nb, 8, 32] -> d, dmin (f16), sc, mn (uint8 [nb,8]), L (uint8 [nb,8,32])."""
    lo = np.minimum(x.min(axis=-1), 0.0)
    hi = np.maximum(x.max(axis=-1), 0.0)
    scale = (hi - lo) / levels
    mins = -lo
    d = _f16(scale.max(axis=-1) / 63.0)
    dmin = _f16(mins.max(axis=-1) / 63.0)
    df = d.astype(np.float32)[..., None]
    dmf = dmin.astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(df > 0, np.rint(scale / df), 0).clip(0, 63).astype(np.uint8)
        mn = np.where(dmf > 0, np.rint(mins / dmf), 0).clip(0, 63).astype(np.uint8)
        es = (df * sc)[..., None]
        em = (dmf * mn)[..., None]
        L = np.where(es > 0, np.rint((x + em) / es), 0).clip(0, levels).astype(np.uint8)
    return d, dmin, sc, mn, L


def quantize_q4_K(x):
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(-1, 8, 32)
    d, dmin, sc, mn, L = _kquant_affine(xb, 15)
    nb = xb.shape[0]
    out = np.zeros((nb, 144), dtype=np.uint8)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    out[:, 2:4] = dmin.view(np.uint8).reshape(nb, 2)
    out[:, 4:16] = _pack_scales_k4(sc, mn)
    Lp = L.reshape(nb, 4, 2, 32)
    out[:, 16:144] = (Lp[:, :, 0, :] | (Lp[:, :, 1, :] << 4)).reshape(nb, 128)
    return out.reshape(lead + (-1,))


def quantize_q5_K(x):
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(-1, 8, 32)
    d, dmin, sc, mn, L = _kquant_affine(xb, 31)
    nb = xb.shape[0]
    out = np.zeros((nb, 176), dtype=np.uint8)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    out[:, 2:4] = dmin.view(np.uint8).reshape(nb, 2)
    out[:, 4:16] = _pack_scales_k4(sc, mn)
    Lp = L.reshape(nb, 4, 2, 32)
    hb = (Lp >> 4).astype(np.uint8)  # [nb, 4, 2, 32] high bits
    qh = np.zeros((nb, 32), dtype=np.uint8)
    for c in range(4):
        qh |= (hb[:, c, 0, :] << (2 * c)) | (hb[:, c, 1, :] << (2 * c + 1))
    out[:, 16:48] = qh
    lo = Lp & 0xF
    out[:, 48:176] = (lo[:, :, 0, :] | (lo[:, :, 1, :] << 4)).reshape(nb, 128)
    return out.reshape(lead + (-1,))


def quantize_q6_K(x):
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(-1, 16, 16)
    nb = xb.shape[0]
    amax = np.abs(xb).max(axis=-1)
    s = amax / 31.0
    d = _f16(s.max(axis=-1) / 127.0)
    df = d.astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(df > 0, np.rint(s / df), 0).clip(0, 127).astype(np.int8)
        es = (df * sc.astype(np.float32))[..., None]
        q = np.where(es > 0, np.rint(xb / es), 0).clip(-32, 31).astype(np.int32)
    L = (q + 32).astype(np.uint8).reshape(nb, 2, 4, 32)  # [half][group of 32 within 128][l]
    out = np.zeros((nb, 210), dtype=np.uint8)
    ql = np.zeros((nb, 2, 64), dtype=np.uint8)
    ql[:, :, 0:32] = (L[:, :, 0, :] & 0xF) | ((L[:, :, 2, :] & 0xF) << 4)
    ql[:, :, 32:64] = (L[:, :, 1, :] & 0xF) | ((L[:, :, 3, :] & 0xF) << 4)
    qh = (L[:, :, 0, :] >> 4) | ((L[:, :, 1, :] >> 4) << 2) | ((L[:, :, 2, :] >> 4) << 4) | ((L[:, :, 3, :] >> 4) << 6)
    out[:, 0:128] = ql.reshape(nb, 128)
    out[:, 128:192] = qh.reshape(nb, 64)
    out[:, 192:208] = sc.view(np.uint8)
    out[:, 208:210] = d.view(np.uint8).reshape(nb, 2)
    return out.reshape(lead + (-1,))


def quantize_q8_0(x):
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(-1, 32)
    nb = xb.shape[0]
    d = _f16(np.abs(xb).max(axis=-1) / 127.0)
    df = d.astype(np.float32)[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(df > 0, np.rint(xb / df), 0).clip(-127, 127).astype(np.int8)
    out = np.zeros((nb, 34), dtype=np.uint8)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    out[:, 2:34] = q.view(np.uint8)
    return out.reshape(lead + (-1,))


def quantize_q4_0(x):
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(-1, 32)
    nb = xb.shape[0]
    idx = np.abs(xb).argmax(axis=-1)
    mx = xb[np.arange(nb), idx]
    d = _f16(mx / -8.0)
    df = d.astype(np.float32)[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(df != 0, np.floor(xb / df + 8.5), 8).clip(0, 15).astype(np.uint8)
    out = np.zeros((nb, 18), dtype=np.uint8)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    out[:, 2:18] = q[:, 0:16] | (q[:, 16:32] << 4)
    return out.reshape(lead + (-1,))


def _pack5(q, nb):
    """q uint8 [nb, 32] in 0..31 -> (qh [nb, 4] bit e = bit 4 of element e, qs [nb, 16] nibbles: element e | element e + 16 << 4)."""
    hb = (q >> 4).astype(np.uint32)
    qh = (hb << np.arange(32, dtype=np.uint32)[None, :]).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    lo = q & 0xF
    return qh.view(np.uint8).reshape(nb, 4), (lo[:, 0:16] | (lo[:, 16:32] << 4)).astype(np.uint8)


def quantize_q4_1(x):
    """block_q4_1 (reference ggml.c:895-900): d, m fp16; q = round((x - min) / d) in 0..15 (quantize_row_q4_1_reference ggml.c:1001-1040)."""
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(-1, 32)
    nb = xb.shape[0]
    mn, mx = xb.min(axis=-1), xb.max(axis=-1)
    d = ((mx - mn) / 15.0).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(d[:, None] != 0, np.floor((xb - mn[:, None]) / d[:, None] + 0.5), 0).clip(0, 15).astype(np.uint8)
    out = np.zeros((nb, 20), dtype=np.uint8)
    out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
    out[:, 2:4] = _f16(mn).view(np.uint8).reshape(nb, 2)
    out[:, 4:20] = q[:, 0:16] | (q[:, 16:32] << 4)
    return out.reshape(lead + (-1,))


def quantize_q5_0(x):
    """block_q5_0 (reference ggml.c:903-908): d fp16 | qh | 16 nibble bytes; q = x / d + 16.5 in 0..31 (ggml.c:1042-1085)."""
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(-1, 32)
    nb = xb.shape[0]
    idx = np.abs(xb).argmax(axis=-1)
    mx = xb[np.arange(nb), idx]
    d = (mx / -16.0).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(d[:, None] != 0, np.floor(xb / d[:, None] + 16.5), 16).clip(0, 31).astype(np.uint8)
    out = np.zeros((nb, 22), dtype=np.uint8)
    out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
    out[:, 2:6], out[:, 6:22] = _pack5(q, nb)
    return out.reshape(lead + (-1,))


def quantize_q5_1(x):
    """block_q5_1 (reference ggml.c:911-917): d, m fp16 | qh | 16 nibble bytes; q = round((x - min) / d) in 0..31 (ggml.c:1087-1130)."""
    x = np.asarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(-1, 32)
    nb = xb.shape[0]
    mn, mx = xb.min(axis=-1), xb.max(axis=-1)
    d = ((mx - mn) / 31.0).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(d[:, None] != 0, np.floor((xb - mn[:, None]) / d[:, None] + 0.5), 0).clip(0, 31).astype(np.uint8)
    out = np.zeros((nb, 24), dtype=np.uint8)
    out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
    out[:, 2:4] = _f16(mn).view(np.uint8).reshape(nb, 2)
    out[:, 4:8], out[:, 8:24] = _pack5(q, nb)
    return out.reshape(lead + (-1,))


def fuzz_blocks(n_blocks, ggml_type, sigma, rng):
    """ARBITRARY block bytes of a quantized type (what a file MAY hold, not what a quantizer would write): every quant / 6-bit scale / min /
    high-bit pattern uniformly random (Q8_0 quants incl. -128, Q6_K scales over the whole int8 range), block scales d / dmin = finite fp16 values
    of either sign around the magnitude that keeps a row's dot product near `sigma` x the activation norm, with zeros, fp16 sub-normals
    and the largest value of the range mixed in.  -> uint8 [n_blocks, block_bytes]"""
    be, bb = G.TYPE_BLOCK[ggml_type]
    raw = rng.integers(0, 256, size=(n_blocks, bb), dtype=np.uint8)
    # typical |weight| of a block with d = 1: Q4_K 63 * 15 / 4, Q5_K 63 * 31 / 4, Q6_K 64 * 16, Q8_0 64, Q4_0 / Q4_1 4, Q5_x 8
    unit = {G.Q4_K: 240.0, G.Q5_K: 490.0, G.Q6_K: 1024.0, G.Q8_0: 64.0, G.Q4_0: 4.0, G.Q4_1: 8.0, G.Q5_0: 8.0, G.Q5_1: 16.0}[ggml_type]

    def scales(n):
        mag = np.float32(sigma) / np.float32(unit) * np.exp2(rng.uniform(-3.0, 1.0, size=n)).astype(np.float32)
        d = (mag * rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=n)).astype(np.float16)
        kind = rng.integers(0, 16, size=n)
        d[kind == 0] = np.float16(0.0)
        d[kind == 1] = np.float16(-0.0)
        sub = np.frombuffer(rng.integers(1, 0x400, size=n, dtype=np.uint16).tobytes(), dtype=np.float16)   # fp16 sub-normals
        d = np.where(kind == 2, sub, d)
        d = np.where(kind == 3, -sub, d)
        return np.ascontiguousarray(d).view(np.uint8).reshape(n, 2)

    if ggml_type in (G.Q4_K, G.Q5_K):
        raw[:, 0:2], raw[:, 2:4] = scales(n_blocks), scales(n_blocks)
    elif ggml_type == G.Q6_K:
        raw[:, 208:210] = scales(n_blocks)
    elif ggml_type in (G.Q8_0, G.Q4_0, G.Q5_0):
        raw[:, 0:2] = scales(n_blocks)
    else:   # Q4_1 / Q5_1: d, m
        raw[:, 0:2], raw[:, 2:4] = scales(n_blocks), scales(n_blocks)
    return raw


def quantize(x, ggml_type, quantizer=None):
    """quantizer=None / "numpy": this file's block quantizers; "reference": the reference's own `ggml_quantize_chunk` through the
    oracle build (oracle/ref.py:quantize_chunk) — the blocks real model files hold, e.g. Q6_K with negative scales and d; "fuzz":
    `fuzz_blocks` (the values of x only seed the generator and set the magnitude)."""
    if quantizer == "reference" and ggml_type not in (G.F32, G.F16):
        from oracle import ref
        return ref.quantize_chunk(x, ggml_type)
    if quantizer == "fuzz" and ggml_type not in (G.F32, G.F16):
        x = np.asarray(x, dtype=np.float32)
        be, bb = G.TYPE_BLOCK[ggml_type]
        rng = np.random.default_rng(int(np.abs(x.reshape(-1)[:64]).view(np.uint32).sum()) & 0x7fffffff)
        sigma = float(np.sqrt(np.mean(np.square(x.reshape(-1)[:1 << 16], dtype=np.float64)))) or 1.0
        return fuzz_blocks(x.size // be, ggml_type, sigma, rng).reshape(x.shape[:-1] + (x.shape[-1] // be * bb,))
    if ggml_type == G.F32:
        return np.ascontiguousarray(x, dtype=np.float32).view(np.uint8).reshape(x.shape[:-1] + (-1,))
    if ggml_type == G.F16:
        return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).view(np.uint8).reshape(x.shape[:-1] + (-1,))
    return {G.Q4_K: quantize_q4_K, G.Q5_K: quantize_q5_K, G.Q6_K: quantize_q6_K, G.Q8_0: quantize_q8_0,
            G.Q4_0: quantize_q4_0, G.Q4_1: quantize_q4_1, G.Q5_0: quantize_q5_0, G.Q5_1: quantize_q5_1}[ggml_type](x)


def dequantize(raw, ggml_type, K):
    """raw: uint8 [..., row_bytes] -> float32 [..., K]; same arithmetic order as the reference dequantize_row_*."""
    raw = np.asarray(raw, dtype=np.uint8)
    lead = raw.shape[:-1]
    f32 = np.float32
    if ggml_type == G.F32:
        return raw.view(np.float32).reshape(lead + (K,))
    if ggml_type == G.F16:
        return raw.view(np.float16).astype(f32).reshape(lead + (K,))
    be, bb = G.TYPE_BLOCK[ggml_type]
    b = raw.reshape(-1, bb)
    nb = b.shape[0]
    if ggml_type == G.Q4_K or ggml_type == G.Q5_K:
        d = b[:, 0:2].copy().view(np.float16).astype(f32).reshape(nb, 1)
        dmin = b[:, 2:4].copy().view(np.float16).astype(f32).reshape(nb, 1)
        sc, mn = _unpack_scales_k4(b[:, 4:16])
        if ggml_type == G.Q4_K:
            qs = b[:, 16:144].reshape(nb, 4, 32)
            L = np.stack([qs & 0xF, qs >> 4], axis=2).astype(f32)  # [nb,4,2,32]
        else:
            qh = b[:, 16:48]
            qs = b[:, 48:176].reshape(nb, 4, 32)
            L = np.zeros((nb, 4, 2, 32), dtype=f32)
            for c in range(4):
                L[:, c, 0, :] = (qs[:, c, :] & 0xF) + (((qh >> (2 * c)) & 1) * 16)
                L[:, c, 1, :] = (qs[:, c, :] >> 4) + (((qh >> (2 * c + 1)) & 1) * 16)
        L = L.reshape(nb, 8, 32)
        d1 = (d * sc.astype(f32))[..., None].astype(f32)
        m1 = (dmin * mn.astype(f32))[..., None].astype(f32)
        y = (d1 * L).astype(f32) - m1
        return y.reshape(lead + (K,)).astype(f32)
    if ggml_type == G.Q6_K:
        ql = b[:, 0:128].reshape(nb, 2, 64)
        qh = b[:, 128:192].reshape(nb, 2, 32)
        sc = b[:, 192:208].copy().view(np.int8).astype(f32).reshape(nb, 2, 8)
        d = b[:, 208:210].copy().view(np.float16).astype(f32).reshape(nb)
        q = np.zeros((nb, 2, 4, 32), dtype=np.int32)
        q[:, :, 0, :] = (ql[:, :, 0:32] & 0xF) | (((qh >> 0) & 3) << 4)
        q[:, :, 1, :] = (ql[:, :, 32:64] & 0xF) | (((qh >> 2) & 3) << 4)
        q[:, :, 2, :] = (ql[:, :, 0:32] >> 4) | (((qh >> 4) & 3) << 4)
        q[:, :, 3, :] = (ql[:, :, 32:64] >> 4) | (((qh >> 6) & 3) << 4)
        q = (q - 32).astype(f32).reshape(nb, 2, 4, 2, 16)
        scr = sc.reshape(nb, 2, 4, 2)  # sc[is + 2*g], is = l/16
        # reference: y = d * sc * q  (left to right)
        y = ((d[:, None, None, None] * scr).astype(f32)[..., None] * q).astype(f32)
        return y.reshape(lead + (K,))
    if ggml_type == G.Q8_0:
        d = b[:, 0:2].copy().view(np.float16).astype(f32).reshape(nb)
        q = b[:, 2:34].copy().view(np.int8).astype(f32)
        return (q * d[:, None]).astype(f32).reshape(lead + (K,))
    if ggml_type == G.Q4_0:
        d = b[:, 0:2].copy().view(np.float16).astype(f32).reshape(nb)
        qs = b[:, 2:18]
        q = np.concatenate([(qs & 0xF).astype(np.int32) - 8, (qs >> 4).astype(np.int32) - 8], axis=1).astype(f32)
        return (q * d[:, None]).astype(f32).reshape(lead + (K,))
    if ggml_type in (G.Q4_1, G.Q5_0, G.Q5_1):   # reference ggml.c:1538-1610: x0*d + m is one fused multiply-add in the reference build
        hdr = 2 if ggml_type == G.Q5_0 else 4
        d = b[:, 0:2].copy().view(np.float16).astype(np.float64).reshape(nb)
        m = b[:, 2:4].copy().view(np.float16).astype(np.float64).reshape(nb) if ggml_type != G.Q5_0 else np.zeros(nb)
        qs = b[:, hdr + (0 if ggml_type == G.Q4_1 else 4):]
        q = np.concatenate([qs & 0xF, qs >> 4], axis=1).astype(np.int32)
        if ggml_type != G.Q4_1:
            qh = b[:, hdr:hdr + 4].copy().view(np.uint32).reshape(nb, 1)
            q |= (((qh >> np.arange(32, dtype=np.uint32)[None, :]) & 1) << 4).astype(np.int32)
        if ggml_type == G.Q5_0:
            q -= 16
        return (q.astype(np.float64) * d[:, None] + m[:, None]).astype(f32).reshape(lead + (K,))   # exact in f64, rounded once = fma
    raise ValueError(ggml_type)


# ----------------------------------------------------------------------------------------------------------------------
# architectures / file type mixes
# ----------------------------------------------------------------------------------------------------------------------

LLAMA_SHAPES = {
    # name: n_vocab, n_embd, n_head, n_head_kv, n_layer, n_ff
    "llama-2-7b": dict(n_vocab=32000, n_embd=4096, n_head=32, n_head_kv=32, n_layer=32, n_ff=11008),
    "llama-2-70b": dict(n_vocab=32000, n_embd=8192, n_head=64, n_head_kv=8, n_layer=80, n_ff=28672),
    # small shapes for parity tests (K-quants need K % 256 == 0)
    "llama-tiny": dict(n_vocab=512, n_embd=256, n_head=4, n_head_kv=2, n_layer=2, n_ff=512),
    "llama-small": dict(n_vocab=1024, n_embd=512, n_head=8, n_head_kv=8, n_layer=3, n_ff=1280),
    # two real-width 7B layers (per-layer kernels at their true shapes, cheap on the CPU side)
    "llama-7b-2l": dict(n_vocab=32000, n_embd=4096, n_head=32, n_head_kv=32, n_layer=2, n_ff=11008),
    # two layers at the real Llama-2-70B widths (GQA 64/8, K = 8192 / 28672)
    "llama-70b-2l": dict(n_vocab=32000, n_embd=8192, n_head=64, n_head_kv=8, n_layer=2, n_ff=28672),
}


def use_more_bits(i, n):
    """reference models/ggml/llama.cpp:4723-4725"""
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def llama_tensor_types(ftype, n_layer):
    """name -> ggml type for every 2-D tensor of a llama GGUF under an ftype (reference llama.cpp:4785-4850)."""
    base = {"Q4_K_M": G.Q4_K, "Q5_K_M": G.Q5_K, "Q8_0": G.Q8_0, "Q4_0": G.Q4_0, "Q4_K_S": G.Q4_K, "Q6_K": G.Q6_K, "F16": G.F16, "F32": G.F32, "Q4_1": G.Q4_1,
            "Q5_0": G.Q5_0, "Q5_1": G.Q5_1}[ftype]
    t = {"token_embd.weight": base, "output.weight": G.Q6_K if ftype in ("Q4_K_M", "Q5_K_M", "Q4_K_S", "Q6_K") else base}
    if ftype in ("Q4_0", "Q4_1", "Q5_0", "Q5_1"):
        t["output.weight"] = G.Q6_K  # reference llama.cpp:4787-4790 (non-falcon: output is always Q6_K when k-quants on)
    for i in range(n_layer):
        more = ftype in ("Q4_K_M", "Q5_K_M") and use_more_bits(i, n_layer)
        for nm in ("attn_q", "attn_k", "attn_output", "ffn_gate", "ffn_up"):
            t["blk.%d.%s.weight" % (i, nm)] = base
        t["blk.%d.attn_v.weight" % i] = G.Q6_K if more else base
        t["blk.%d.ffn_down.weight" % i] = G.Q6_K if more else base
        if ftype == "Q4_K_S" and i < 4:   # llama.cpp:4801 (attn_v) and :4830-4832 (ffn_down): Q5_K in the first four layers
            t["blk.%d.attn_v.weight" % i] = G.Q5_K
            t["blk.%d.ffn_down.weight" % i] = G.Q5_K
    return t


def make_vocab(n_vocab):
    toks = [b"<unk>", b"<s>", b"</s>"] + [b"<0x%02X>" % i for i in range(256)]
    types = [2, 3, 3] + [6] * 256
    i = 0
    while len(toks) < n_vocab:
        toks.append(("▁t%d" % i).encode("utf-8"))
        types.append(1)
        i += 1
    scores = [0.0] * 259 + [-float(j + 1) for j in range(len(toks) - 259)]
    return toks[:n_vocab], scores[:n_vocab], types[:n_vocab]


class BlockPool:
    """Pre-quantized pool of blocks of one ggml type, drawn from N(0, sigma); tensors are assembled by sampling
    blocks from the pool (fast path for multi-GB synthetic models; every block is a legal quantization of real
    Gaussian data, every tensor is a different random sequence of them)."""

    def __init__(self, ggml_type, sigma, rng, n_blocks=1 << 13, quantizer=None):
        be, bb = G.TYPE_BLOCK[ggml_type]
        x = rng.standard_normal((n_blocks, be), dtype=np.float32) * np.float32(sigma)
        self.blocks = quantize(x, ggml_type, quantizer).reshape(n_blocks, bb)
        self.n, self.bb, self.be = n_blocks, bb, be

    def draw(self, rng, n_blocks):
        idx = rng.integers(0, self.n, size=n_blocks, dtype=np.int32)
        return self.blocks[idx]


class _WeightSource:
    def __init__(self, seed, pooled, quantizer=None):
        self.rng = np.random.default_rng(seed)
        self.pooled = pooled
        self.quantizer = quantizer
        self.pools = {}

    def matrix(self, rows, K, ggml_type, sigma):
        """-> uint8 [rows * row_bytes]"""
        if ggml_type in (G.F32, G.F16) or not self.pooled:
            x = self.rng.standard_normal((rows, K), dtype=np.float32) * np.float32(sigma)
            return quantize(x, ggml_type, self.quantizer).reshape(-1)
        key = (ggml_type, round(float(sigma), 9))
        if key not in self.pools:
            self.pools[key] = BlockPool(ggml_type, sigma, self.rng, quantizer=self.quantizer)
        pool = self.pools[key]
        return pool.draw(self.rng, rows * (K // pool.be)).reshape(-1)

    def norm(self, n):
        return (1.0 + 0.1 * self.rng.standard_normal(n, dtype=np.float32)).astype(np.float32)


def make_spm_vocab(n_vocab=512):
    """A small sentencepiece-style vocabulary WITH real merge structure (the default `make_vocab` has only byte tokens and opaque
    pieces): control and byte tokens, then pieces whose scores fall with their index the way trained SPM vocabularies do, so the
    reference's score-ordered bigram merging (llama.cpp:3080-3210) has choices to make — "▁hello" exists as a piece, and so do the
    intermediate pieces its merges pass through."""
    toks = [b"<unk>", b"<s>", b"</s>"] + [b"<0x%02X>" % i for i in range(256)]
    types = [2, 3, 3] + [6] * 256
    words = ["▁", "e", "t", "a", "o", "i", "n", "s", "h", "r", "l", "d", "u", "w", "m", "c", "f", "g", "y", "p", "b", "v", "k", ",", ".", "!",
             "▁t", "he", "▁a", "in", "▁the", "er", "▁s", "re", "on", "▁w", "at", "en", "nd", "▁o", "or", "▁c", "es", "is", "it", "an", "▁b",
             "ing", "ed", "ou", "▁h", "ar", "ll", "▁he", "▁hel", "lo", "▁hell", "▁hello", "wor", "ld", "▁wor", "▁world", "▁an", "▁and", "▁in",
             "▁th", "▁to", "▁of", "al", "le", "ion", "▁f", "▁m", "as", "▁is", "st", "▁p", "▁d", "ic", "▁it", "▁that", "th", "at", "▁wh",
             "▁whe", "▁when", "ess", "▁you", "▁for", "ent", "ly", "▁be", "▁on", "▁we", "ver", "▁ha", "▁re", "ld", "gh", "ght", "▁▁", "▁▁▁",
             "é", "ï", "na", "ve", "ca", "fé", "ca", "12", "34", "1", "2", "3", "4", "5"]
    seen = set(toks)
    for w in words:
        b = w.encode("utf-8")
        if b not in seen:
            seen.add(b)
            toks.append(b)
            types.append(1)
    i = 0
    while len(toks) < n_vocab:
        toks.append(("▁x%d" % i).encode("utf-8"))
        types.append(1)
        i += 1
    scores = [0.0] * 259 + [-float(j + 1) for j in range(len(toks) - 259)]
    return toks[:n_vocab], scores[:n_vocab], types[:n_vocab]


def write_llama_gguf(path, shape="llama-2-7b", ftype="Q4_K_M", seed=1234, n_ctx_train=4096, pooled=None,
                     rope_freq_base=None, rms_eps=1e-5, overrides=None, vocab=None, type_overrides=None, quantizer=None, tensor_data=None):
    """Write a synthetic llama-architecture GGUF v2 file.  Returns the hparams dict.
    tensor_data: {tensor name: float32 array} — these tensors hold the given values (quantized to the tensor's type) instead of random ones.
    type_overrides: {tensor name or dotted name suffix: ggml type} applied on top of the ftype's mix (e.g. {"attn_v.weight": G.Q8_0};
    "output.weight" names the head only, not blk.N.attn_output.weight).  quantizer: see `quantize`."""
    hp = dict(LLAMA_SHAPES[shape]) if isinstance(shape, str) else dict(shape)
    if overrides:
        hp.update(overrides)
    n_vocab, n_embd, n_head, n_head_kv = hp["n_vocab"], hp["n_embd"], hp["n_head"], hp["n_head_kv"]
    n_layer, n_ff = hp["n_layer"], hp["n_ff"]
    head_dim = n_embd // n_head
    n_embd_gqa = head_dim * n_head_kv
    if pooled is None:
        pooled = n_embd >= 2048
    src = _WeightSource(seed, pooled, quantizer)
    types = llama_tensor_types(ftype, n_layer)
    for suffix, ty in (type_overrides or {}).items():
        for name in list(types):
            if name == suffix or name.endswith("." + suffix):
                types[name] = ty

    w = G.GGUFWriter(path)
    w.add_str("general.architecture", "llama")
    w.add_str("general.name", "synthetic-%s-%s" % (shape if isinstance(shape, str) else "custom", ftype))
    w.add_u32("llama.context_length", n_ctx_train)
    w.add_u32("llama.embedding_length", n_embd)
    w.add_u32("llama.block_count", n_layer)
    w.add_u32("llama.feed_forward_length", n_ff)
    w.add_u32("llama.rope.dimension_count", head_dim)
    w.add_u32("llama.attention.head_count", n_head)
    w.add_u32("llama.attention.head_count_kv", n_head_kv)
    w.add_f32("llama.attention.layer_norm_rms_epsilon", rms_eps)
    if rope_freq_base is not None:
        w.add_f32("llama.rope.freq_base", rope_freq_base)
    toks, scores, ttypes = vocab if vocab is not None else make_vocab(n_vocab)
    w.add_str("tokenizer.ggml.model", "llama")
    w.add_arr("tokenizer.ggml.tokens", G.T_STR, toks)
    w.add_arr("tokenizer.ggml.scores", G.T_F32, scores)
    w.add_arr("tokenizer.ggml.token_type", G.T_I32, ttypes)
    w.add_u32("tokenizer.ggml.bos_token_id", 1)
    w.add_u32("tokenizer.ggml.eos_token_id", 2)
    w.add_u32("tokenizer.ggml.unknown_token_id", 0)

    given = tensor_data or {}

    def mat(name, rows, K, sigma):
        t = types[name]
        if name in given:
            x = np.ascontiguousarray(given[name], dtype=np.float32).reshape(rows, K)
            w.add_tensor(name, (K, rows), t, lambda: quantize(x, t, quantizer).reshape(-1))
        else:
            w.add_tensor(name, (K, rows), t, lambda: src.matrix(rows, K, t, sigma))

    def vec(name, n):
        if name in given:
            v = np.ascontiguousarray(given[name], dtype=np.float32).reshape(n)
            w.add_tensor(name, (n,), G.F32, lambda: v.view(np.uint8))
        else:
            w.add_tensor(name, (n,), G.F32, lambda: src.norm(n).view(np.uint8))

    s_e = 1.0 / np.sqrt(n_embd)
    s_f = 1.0 / np.sqrt(n_ff)
    mat("token_embd.weight", n_vocab, n_embd, 1.0)
    for i in range(n_layer):
        p = "blk.%d." % i
        vec(p + "attn_norm.weight", n_embd)
        mat(p + "attn_q.weight", n_embd, n_embd, s_e)
        mat(p + "attn_k.weight", n_embd_gqa, n_embd, s_e)
        mat(p + "attn_v.weight", n_embd_gqa, n_embd, s_e)
        mat(p + "attn_output.weight", n_embd, n_embd, s_e)
        vec(p + "ffn_norm.weight", n_embd)
        mat(p + "ffn_gate.weight", n_ff, n_embd, s_e)
        mat(p + "ffn_down.weight", n_embd, n_ff, s_f)
        mat(p + "ffn_up.weight", n_ff, n_embd, s_e)
    vec("output_norm.weight", n_embd)
    mat("output.weight", n_vocab, n_embd, s_e)
    w.write()
    hp.update(head_dim=head_dim, n_embd_gqa=n_embd_gqa, ftype=ftype, rms_eps=rms_eps)
    return hp


def prompt_tokens(n, n_vocab=32000):
    """The benchmark prompt of BASELINE.md §3: [BOS] + [259 + (7*i mod 3000)], clipped into the vocab."""
    span = min(3000, n_vocab - 259)
    return [1] + [259 + (7 * i) % span for i in range(n - 1)]


def weight_bytes_per_token(path):
    """Algorithmic weight bytes one decoded token must read: every 2-D tensor except token_embd (one row of it)."""
    f = G.GGUFFile(path)
    total = 0
    for name, (shape, t, data) in f.tensors.items():
        if name == "token_embd.weight":
            total += G.row_bytes(t, shape[0])
        else:
            total += data.size
    return total


# ---------------------------------------------------------------------------------------------------------------------
# Falcon (reference llm_build_falcon, models/ggml/llama.cpp:2493-2798)
# ---------------------------------------------------------------------------------------------------------------------
FALCON_SHAPES = {
    # n_ff is 4*n_embd in the real models; the "40b" style has the second attention norm (attn_norm_2)
    "falcon-40b": dict(n_vocab=65024, n_embd=8192, n_head=128, n_head_kv=8, n_layer=60, n_ff=32768, norm2=True),
    "falcon-7b": dict(n_vocab=65024, n_embd=4544, n_head=71, n_head_kv=1, n_layer=32, n_ff=18176, norm2=False),
    # parity-test shapes (head_dim 64 like the real models; K-quants need K % 256 == 0)
    "falcon-tiny": dict(n_vocab=512, n_embd=256, n_head=4, n_head_kv=2, n_layer=2, n_ff=1024, norm2=True),
    "falcon-tiny7": dict(n_vocab=512, n_embd=256, n_head=4, n_head_kv=1, n_layer=2, n_ff=1024, norm2=False),
    "falcon-small": dict(n_vocab=1024, n_embd=1024, n_head=16, n_head_kv=2, n_layer=3, n_ff=4096, norm2=True),
    # two layers at the real Falcon-7B widths (4544 = 71 heads of 64: rows of 142 32-blocks; K-quant files of this model fall back to
    # the 32-block types for every tensor with such rows, llama.cpp:4850-4870)
    "falcon-7b-2l": dict(n_vocab=65024, n_embd=4544, n_head=71, n_head_kv=1, n_layer=2, n_ff=18176, norm2=False),
    # two layers at the real Falcon-40B widths (K = 8192 / 32768, 65024 x 8192 Q8_0 head)
    "falcon-40b-2l": dict(n_vocab=65024, n_embd=8192, n_head=128, n_head_kv=8, n_layer=2, n_ff=32768, norm2=True),
}


def falcon_tensor_types(ftype, n_layer):
    """name -> ggml type of the 2-D tensors of a falcon GGUF (reference llama.cpp:4785-4850, arch == FALCON rules)."""
    base = {"Q4_K_M": G.Q4_K, "Q5_K_M": G.Q5_K, "Q8_0": G.Q8_0, "Q4_0": G.Q4_0, "Q4_1": G.Q4_1, "Q5_0": G.Q5_0, "Q5_1": G.Q5_1, "F16": G.F16, "F32": G.F32}[ftype]
    t = {"token_embd.weight": base, "output.weight": base if ftype in ("F16", "F32") else G.Q8_0}   # :4787-4788: a quantized falcon output is always Q8_0
    for i in range(n_layer):
        qkv, down = base, base
        if ftype == "Q4_K_M":
            qkv = G.Q5_K                                                          # :4845
            down = G.Q6_K if i < 2 else (G.Q5_K if use_more_bits(i, n_layer) else G.Q4_K)   # :4822-4824
        elif ftype == "Q5_K_M":
            qkv = G.Q6_K                                                          # :4846
            down = G.Q6_K if use_more_bits(i, n_layer) else base                   # :4830
        t["blk.%d.attn_qkv.weight" % i] = qkv
        t["blk.%d.attn_output.weight" % i] = base
        t["blk.%d.ffn_up.weight" % i] = base
        t["blk.%d.ffn_down.weight" % i] = down
    return t


def make_bpe_vocab(n_vocab):
    """A byte-level BPE vocabulary for the reference's `llm_tokenizer_bpe` (llama.cpp:3228-3388): that tokenizer looks
    symbols up as RAW strings (no GPT-2 byte->unicode remap in this version) and falls back to single-byte tokens, and
    the loader tokenizes "\n" to find the linefeed id (:1751) — so all 256 single-byte strings are tokens.  Token 11 is
    <|endoftext|> like Falcon's (bos = eos = 11 are hard-coded, :1719-1720).  Then synthetic two-letter merges."""
    toks = [bytes([b]) for b in range(256)]
    toks[11] = b"<|endoftext|>"
    merges = []
    i = 0
    while len(toks) < n_vocab:
        a, b = chr(ord("a") + (i % 26)), chr(ord("a") + ((i // 26) % 26))
        if i < 676:
            piece = a + b
            merges.append(a + " " + b)
        else:
            piece = " " + a + b + str(i)
        toks.append(piece.encode("ascii"))
        i += 1
    return toks[:n_vocab], merges


def write_falcon_gguf(path, shape="falcon-tiny", ftype="Q4_K_M", seed=1234, n_ctx_train=2048, pooled=None, norm_eps=1e-5,
                      overrides=None, quantizer=None):
    """Write a synthetic falcon-architecture GGUF v2 file.  Returns the hparams dict."""
    hp = dict(FALCON_SHAPES[shape]) if isinstance(shape, str) else dict(shape)
    if overrides:
        hp.update(overrides)
    n_vocab, n_embd, n_head, n_head_kv = hp["n_vocab"], hp["n_embd"], hp["n_head"], hp["n_head_kv"]
    n_layer, n_ff = hp["n_layer"], hp["n_ff"]
    head_dim = n_embd // n_head
    if pooled is None:
        pooled = n_embd >= 2048
    src = _WeightSource(seed, pooled, quantizer)
    types = falcon_tensor_types(ftype, n_layer)
    w = G.GGUFWriter(path)
    w.add_str("general.architecture", "falcon")
    w.add_str("general.name", "synthetic-%s-%s" % (shape if isinstance(shape, str) else "custom", ftype))
    w.add_u32("falcon.context_length", n_ctx_train)
    w.add_u32("falcon.embedding_length", n_embd)
    w.add_u32("falcon.block_count", n_layer)
    w.add_u32("falcon.feed_forward_length", n_ff)
    w.add_u32("falcon.attention.head_count", n_head)
    w.add_u32("falcon.attention.head_count_kv", n_head_kv)
    w.add_f32("falcon.attention.layer_norm_epsilon", norm_eps)
    toks, merges = make_bpe_vocab(n_vocab)
    w.add_str("tokenizer.ggml.model", "gpt2")
    w.add_arr("tokenizer.ggml.tokens", G.T_STR, toks)
    w.add_arr("tokenizer.ggml.scores", G.T_F32, [0.0] * n_vocab)
    w.add_arr("tokenizer.ggml.token_type", G.T_I32, [1] * n_vocab)
    w.add_arr("tokenizer.ggml.merges", G.T_STR, [m.encode("ascii") for m in merges])
    w.add_u32("tokenizer.ggml.bos_token_id", 11)
    w.add_u32("tokenizer.ggml.eos_token_id", 11)

    def mat(name, rows, K, sigma):
        t = types[name]
        w.add_tensor(name, (K, rows), t, lambda: src.matrix(rows, K, t, sigma))

    def vec(name, n, bias=False):
        if bias:
            w.add_tensor(name, (n,), G.F32, lambda: (src.norm(n) - np.float32(1.0)).astype(np.float32).view(np.uint8))
        else:
            w.add_tensor(name, (n,), G.F32, lambda: src.norm(n).view(np.uint8))

    s_e, s_f = 1.0 / np.sqrt(n_embd), 1.0 / np.sqrt(n_ff)
    mat("token_embd.weight", n_vocab, n_embd, 1.0)
    vec("output_norm.weight", n_embd)
    vec("output_norm.bias", n_embd, bias=True)
    mat("output.weight", n_vocab, n_embd, s_e)
    for i in range(n_layer):
        p = "blk.%d." % i
        vec(p + "attn_norm.weight", n_embd)
        vec(p + "attn_norm.bias", n_embd, bias=True)
        if hp.get("norm2"):
            vec(p + "attn_norm_2.weight", n_embd)
            vec(p + "attn_norm_2.bias", n_embd, bias=True)
        mat(p + "attn_qkv.weight", (n_head + 2 * n_head_kv) * head_dim, n_embd, s_e)
        mat(p + "attn_output.weight", n_embd, n_embd, s_e)
        mat(p + "ffn_up.weight", n_ff, n_embd, s_e)
        mat(p + "ffn_down.weight", n_embd, n_ff, s_f)
    w.write()
    hp.update(dict(head_dim=head_dim, ftype=ftype, norm_eps=norm_eps))
    return hp


# ---------------------------------------------------------------------------------------------------------------------
# GPT-2, legacy (pre-GGUF) GGML file as read by the reference's gpt2_model_load (models/llms/gpt2.cc:61-381)
# ---------------------------------------------------------------------------------------------------------------------
GPT2_SHAPES = {
    "gpt2-117m": dict(n_vocab=50257, n_ctx=1024, n_embd=768, n_head=12, n_layer=12),
    "gpt2-tiny": dict(n_vocab=512, n_ctx=96, n_embd=256, n_head=4, n_layer=2),
    "gpt2-xl-2l": dict(n_vocab=50257, n_ctx=1024, n_embd=1600, n_head=25, n_layer=2),   # GPT-2 XL widths: rows of 50 / 200 blocks (not whole groups of four), 25 heads
    # the reference's starcoder / gptbigcode loader reads the same container (models/llms/starcoder.cc); heads of 64 like the real ones
    "starcoder-tiny": dict(n_vocab=512, n_ctx=96, n_embd=384, n_head=6, n_layer=2),
    "starcoder-1b": dict(n_vocab=49152, n_ctx=8192, n_embd=2048, n_head=16, n_layer=24),
    "starcoder-7b-2l": dict(n_vocab=49152, n_ctx=2048, n_embd=4096, n_head=32, n_layer=2),   # StarCoderBase-7B widths: c_proj rows of 16384
    "starcoder-1b-4l": dict(n_vocab=49152, n_ctx=2048, n_embd=2048, n_head=16, n_layer=4),   # StarCoderBase-1B widths, heads of 128
}

# pieces the reference registers as special for starcoder when the vocabulary holds them (models/llms/starcoder.cc:123-138); the
# synthetic vocabulary carries a subset, so the "only those present" rule is exercised too
STARCODER_PIECES = [b"<|endoftext|>", b"<fim-prefix>", b"<fim-middle>", b"<fim-suffix>", b"<|system|>", b"<|user|>", b"<|assistant|>",
                    b"<|end|>", b"<|", b"|>", b"<fim", b"end"]


def make_gpt2_vocab(n_vocab):
    """Raw-byte pieces: the 256 single bytes, then two-letter and three-letter ASCII pieces (reference gpt_tokenize,
    models/common.cc, looks pieces up as raw strings after its regex split)."""
    toks = [bytes([b]) for b in range(256)]
    i = 0
    while len(toks) < n_vocab:
        a, b, c = chr(ord("a") + (i % 26)), chr(ord("a") + ((i // 26) % 26)), chr(ord("a") + ((i // 676) % 26))
        piece = (a + b) if i < 676 else (" " + a + b + c)
        toks.append(piece.encode("ascii"))
        i += 1
    return toks[:n_vocab]


def write_gpt2_ggml(path, shape="gpt2-tiny", seed=1234, ftype=2, pooled=None, lm_head=False, pieces=None, quantizer=None):
    """Synthetic GPT-2 in the legacy GGML container (magic 0x67676d6c, 6 x i32 hparams, vocab, tensors; ftype 2 = Q4_0,
    stored as ftype + 1000*GGML_QNT_VERSION).  Returns the hparams dict."""
    import struct
    hp = dict(GPT2_SHAPES[shape]) if isinstance(shape, str) else dict(shape)
    V, C, E, H, NL = hp["n_vocab"], hp["n_ctx"], hp["n_embd"], hp["n_head"], hp["n_layer"]
    if pooled is None:
        pooled = E >= 2048
    src = _WeightSource(seed, pooled, quantizer)
    wtype = {0: G.F32, 1: G.F16, 2: G.Q4_0, 3: G.Q4_1, 7: G.Q8_0, 8: G.Q5_0, 9: G.Q5_1}[ftype]   # enum ggml_ftype (ggml.h:322-336)
    rng = np.random.default_rng(seed + 17)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 0x67676d6c))
        f.write(struct.pack("<6i", V, C, E, H, NL, ftype + 1000 * 2))
        toks = make_gpt2_vocab(V)
        if pieces:   # named pieces take the last ids of the vocabulary
            toks[V - len(pieces):] = list(pieces)
        f.write(struct.pack("<i", V))
        for t in toks:
            f.write(struct.pack("<I", len(t)) + t)

        def put(name, dims, ttype, data):
            nb = name.encode("ascii")
            f.write(struct.pack("<3i", len(dims), len(nb), ttype))
            for d in dims:
                f.write(struct.pack("<i", int(d)))
            f.write(nb)
            f.write(np.ascontiguousarray(data).tobytes())

        def mat(name, rows, K, sigma):
            put(name, (K, rows), wtype, src.matrix(rows, K, wtype, sigma))

        def gain(name, n):
            put(name, (n,), G.F32, src.norm(n))

        def bias(name, n, s=0.1):
            put(name, (n,), G.F32, (rng.standard_normal(n) * s).astype(np.float32))

        s_e, s_f = 1.0 / np.sqrt(E), 1.0 / np.sqrt(4 * E)
        gain("model/ln_f/g", E)
        bias("model/ln_f/b", E)
        mat("model/wte", V, E, 0.06)   # small: the tied lm_head then gives a spread-out next-token distribution
        put("model/wpe", (E, C), G.F32, (rng.standard_normal((C, E)) * 0.3).astype(np.float32))
        if lm_head:
            mat("model/lm_head", V, E, s_e)
        for i in range(NL):
            p = "model/h%d/" % i
            gain(p + "ln_1/g", E); bias(p + "ln_1/b", E)
            gain(p + "ln_2/g", E); bias(p + "ln_2/b", E)
            mat(p + "attn/c_attn/w", 3 * E, E, s_e); bias(p + "attn/c_attn/b", 3 * E)
            mat(p + "attn/c_proj/w", E, E, s_e); bias(p + "attn/c_proj/b", E)
            mat(p + "mlp/c_fc/w", 4 * E, E, s_e); bias(p + "mlp/c_fc/b", 4 * E)
            mat(p + "mlp/c_proj/w", E, 4 * E, s_f); bias(p + "mlp/c_proj/b", E)
    hp.update(dict(ftype=ftype))
    return hp


MPT_SHAPES = {
    # head sizes of 64 and 128 (MPT-7B: d_model 4096, 32 heads); a head count that is not a power of two exercises both ALiBi
    # slope branches (ggml.c:12243-12247)
    "mpt-tiny": dict(n_vocab=512, max_seq_len=96, n_embd=384, n_head=6, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.75),
    "mpt-tiny128": dict(n_vocab=512, max_seq_len=2048, n_embd=512, n_head=4, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0),
    "mpt-7b-2l": dict(n_vocab=50432, max_seq_len=2048, n_embd=4096, n_head=32, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0),
    # heads of 112 (MPT-30B: d_model 7168, 64 heads): the K.Q dot is three 32-element steps + ggml_vec_dot_f16's scalar tail of 16
    "mpt-tiny112": dict(n_vocab=512, max_seq_len=96, n_embd=896, n_head=8, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0),
    "mpt-30b-2l": dict(n_vocab=50432, max_seq_len=2048, n_embd=7168, n_head=64, n_layer=2, alibi_bias_max=8.0, clip_qkv=0.0),
}


def write_mpt_ggml(path, shape="mpt-tiny", seed=1234, ftype=2, pooled=None, pieces=None, quantizer=None):
    """Synthetic MPT in the legacy GGML container as the reference's mpt loader reads it (models/llms/mpt.cc:50-363): magic,
    d_model, max_seq_len, n_heads, n_layers, n_vocab, alibi_bias_max (f32), clip_qkv (f32), ftype; the vocabulary without a count,
    pieces in UTF-8 (the loader keeps the low byte of every code point); quantized wte, f32 norm gains, four matrices per layer."""
    import struct
    hp = dict(MPT_SHAPES[shape]) if isinstance(shape, str) else dict(shape)
    V, C, E, H, NL = hp["n_vocab"], hp["max_seq_len"], hp["n_embd"], hp["n_head"], hp["n_layer"]
    if pooled is None:
        pooled = E >= 2048
    src = _WeightSource(seed, pooled, quantizer)
    wtype = {0: G.F32, 1: G.F16, 2: G.Q4_0, 3: G.Q4_1, 7: G.Q8_0, 8: G.Q5_0, 9: G.Q5_1}[ftype]   # enum ggml_ftype (ggml.h:322-336)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 0x67676d6c))
        f.write(struct.pack("<5i2fi", E, C, H, NL, V, hp["alibi_bias_max"], hp["clip_qkv"], ftype + 1000 * 2))
        toks = make_gpt2_vocab(V)
        if pieces:
            toks[V - len(pieces):] = list(pieces)
        for t in toks:
            u = "".join(chr(b) for b in t).encode("utf-8")   # code point b -> the loader's low byte b
            f.write(struct.pack("<I", len(u)) + u)

        def put(name, dims, ttype, data):
            nb = name.encode("ascii")
            f.write(struct.pack("<3i", len(dims), len(nb), ttype))
            for d in dims:
                f.write(struct.pack("<i", int(d)))
            f.write(nb)
            f.write(np.ascontiguousarray(data).tobytes())

        def mat(name, rows, K, sigma):
            put(name, (K, rows), wtype, src.matrix(rows, K, wtype, sigma))

        s_e, s_f = 1.0 / np.sqrt(E), 1.0 / np.sqrt(4 * E)
        mat("transformer.wte.weight", V, E, 0.06)
        put("transformer.norm_f.weight", (E,), G.F32, src.norm(E))
        for i in range(NL):
            p = "transformer.blocks.%d." % i
            put(p + "norm_1.weight", (E,), G.F32, src.norm(E))
            mat(p + "attn.Wqkv.weight", 3 * E, E, s_e)
            mat(p + "attn.out_proj.weight", E, E, s_e)
            put(p + "norm_2.weight", (E,), G.F32, src.norm(E))
            mat(p + "ffn.up_proj.weight", 4 * E, E, s_e)
            mat(p + "ffn.down_proj.weight", E, 4 * E, s_f)
    hp.update(dict(ftype=ftype, n_ctx=C))
    return hp
