// Mat-vec, generation 9 (K-quants, every K): the lane that computes sumi[l] owns chain l.
//
// Generation 7 (deleted in round 3 after the A/B; git history) kept a record's nibble bytes in file order, so a lane's 16 bytes were four AVX lanes of ONE
// 32-element vector pair: every step paid a transpose-reduce over four lanes, a round trip through a wave-private LDS buffer to
// hand the chain operands to the lanes that replay the reference's f32 chain, and a prologue that 4 of 16 waves computed for all
// 256 workgroups (profiles/r02_v7_*: 133 VALU instructions per 1152-byte record, a third of every launch spent before the first
// block math).  Generation 9 keeps the work decomposition — a wave owns a PAIR OF ROWS from its first byte to its epilogue, a step
// is one record of 2 rows x 4 consecutive K-blocks, a register ring of three or four records per wave (matvec_v9_kernel) — and changes what a lane does:
//   * layout LAYOUT_L9 (quant.h): lane (row, l, c) = 32 * row + 4 * l + c of a step holds, in ONE 16-byte load, the four dwords
//     that AVX lane l of the reference touches in block 4 * s + c of its row (all eight 32-element vectors, four elements each).
//     It computes the block's complete integer lane sum sumi[l] locally — 8 x dot4 against the Q8_K image, 8 scale multiplies —
//     with no cross-lane reduction at all.
//   * the four lanes of a quad are the four blocks of a step for one (row, l) chain: three DPP quad broadcasts per operand hand
//     (float)sumi and y.d * d of block c to all four, every lane of the quad performs the reference's four fmas in block order
//     (acc = fma(d_c, s_c, acc); the quad_perm move folds into v_fmac_f32_dpp).  No LDS exchange, no wave_lds_sync.
//   * scales and mins of Q4_K / Q5_K: the record re-encodes the file's 12 bytes of 6-bit fields (same 12 bytes) so that every
//     scale sits inside one word — all lanes extract all eight with one v_bfe each — and lane l extracts the ONE min m_l it needs,
//     multiplies it with the sum of vector l's quants, one DPP add pairs (2t, 2t+1) into prod[t] (Q4_K) / three add up the block
//     total (Q5_K); the min chain runs beside the main one.  (Byte-expanded scales — 148-byte blocks, SDWA byte operands, 15
//     instructions fewer per record — were measured: no faster with one header load, 5 % slower with the three loads an aligned
//     20-byte header takes; the step is not VALU-bound.)
//   * Q6_K's -32 offset costs no instruction: the image carries -32 * (sum of the four quants) per activation word and the dot
//     product starts from it (accumulator operand of v_dot4).
//   * prologue: one WAVE per 256-block, all 16 waves (a lane owns 4 consecutive activations: one float4 of x and one of the norm
//     weights per block — nothing redundant in front of the weight stream), amax / first-maximum by DPP butterflies and one
//     ballot; the Q8_K image is written in the order the block math reads it ([half][l][4 vectors], 80-dword block stride:
//     conflict-free ds_read_b128), with the sums of 32 and y.d in the same 320-byte block record.  Blocks past the row's end
//     (the padding slots of a last record) are zero images with y.d = 0, so the step needs no "how many blocks exist" branch.
// Arithmetic and its order are generation 7's (= the reference build's, kernels_exact.h header): integer lane sums exact, one fma
// per block and lane in block order, hsum_float_8's tree at the end of the row.
#pragma once
#include "kernels_exact.h"

// Stamps inside the prologue cost eight registers the 128-register instantiations do not have: a -DV9_PROTRACE build only
// (make OUT=../lib_trace EXTRA=-DV9_PROTRACE; tools/gpu_trace.py with SITES_LIB).
#ifdef V9_PROTRACE
#define V9_STAMP(cond, dst) do { if (cond) (dst) = clock64_dev(); } while (0)
#else
#define V9_STAMP(cond, dst) do { } while (0)
#endif
constexpr int kImg9Stride = 144;  // dwords per block image: 64 quant words | 8 sums of 32 | y.d | 7 unused | 64 x -32 * (sum of a word's quants)
template <int MAXK> struct Img9 {
    int blk[(MAXK / 256 + 3) * kImg9Stride];   // >= (MAXK / 512 + 1) * kImg9bStride: the 32-block image fits too
    double red[2][MAXK / 256];   // per-BLOCK partial sums of the norm (second array: LayerNorm's second moment)
    unsigned cnt;        // arrivals of the waves that own blocks: only they meet for the norm's sums
};
// 32-element block types (Q8_0 / Q4_0 weights x Q8_0 activations): one image per record step (16 blocks = 512 activations):
// 128 quant words [c][l][t] | 128 words -8 * (sum of a word's quants) [c][l][t] (Q4_0: (nibble - 8) . a = nibble . a - 8 * sum) | 16 y.d [c][t]
constexpr int kImg9bStride = 272;
constexpr int kV9MaxUnits = 32;   // units one wave may own in a launch (its results wait in LDS for the epilogue pass)
template <int MAXK> struct SmemV9 {
    Img9<MAXK> L;
    float RES[16][2 * kV9MaxUnits];   // [wave][2 * unit + row]
};

// The greedy pick, first half, at the end of the head launch (MatvecArgs::pick_ws; reference semantics: llama_sample_top_k with k = 1
// keeps the first maximum, models/llms/llama.cc:53-84).  Every WAVE leaves the first maximum of ITS logits rows as one 64-bit key (kernels.h:
// pick_key) in its own slot [workgroup][wave], with a plain store: no atomics, no barrier, nobody waits for anybody.  pick_cont_kernel
// (kernels.h) reads the 16 x grid slots instead of the n_vocab logits.  (Round 5 measured the one-launch forms first — every workgroup
// merging into one word with device-scope atomics, the last arriver finishing the job: + 6.8 us on the head launch for the atomics of 256
// workgroups on one address, + 7 us for the embedding row behind them; and a workgroup-level key behind a barrier: + 2.2 us.
// profiles/r05_head_fold.txt.)
DEV void v9_pick_store(const MatvecArgs& a, int wv, int lane, float bv, int bi) {
    pick_wave_reduce(bv, bi);
    if (lane == 0) ((unsigned long long*)a.pick_ws)[(int)blockIdx.x * 16 + wv] = pick_key(bv, bi);
}

struct Lane9 {
    int c, l, row, slot;
    uint32_t o16, o8, o4, o_hdr, o_hdr5, o_sc6, o_d6;   // byte offsets inside a record
    uint32_t o_qs40, o_d80, o_d40;                      // Q4_0 quants, Q8_0 / Q4_0 block scales
    int a_b;        // 32-block types: word offset of the lane's four activation words (blocks 4t + c) inside a step image
    int sh40;       // Q4_0: 0 (l < 4: low nibbles) or 4 (high nibbles)
    int a_w;        // word offset of this lane's first four activation words inside a block image
    int m_sh;       // bit offset of min m_l inside the pair W3:W2 of the block header
};
DEV Lane9 lane9(int lane) {
    Lane9 G;
    G.c = lane & 3; G.l = (lane >> 2) & 7; G.row = lane >> 5; G.slot = 4 * G.row + G.c;
    G.o16 = (uint32_t)lane * 16u; G.o8 = 1024u + (uint32_t)lane * 8u; G.o4 = 1024u + (uint32_t)lane * 4u;
    G.o_hdr = 1024u + (uint32_t)G.slot * 16u;
    G.o_hdr5 = 1280u + (uint32_t)G.slot * 16u;
    G.o_sc6 = 1536u + (uint32_t)G.slot * 16u + (uint32_t)(G.l >> 2) * 8u;
    G.o_d6 = 1664u + (uint32_t)G.slot * 2u;
    G.a_w = 4 * G.l;
    G.o_qs40 = (uint32_t)((G.row * 4 + (G.l & 3)) * 4 + G.c) * 16u;
    G.o_d80 = 1024u + (uint32_t)(G.row * 4 + G.c) * 8u;
    G.o_d40 = 512u + (uint32_t)(G.row * 4 + G.c) * 8u;
    G.a_b = (G.c * 8 + G.l) * 4;
    G.sh40 = G.l >= 4 ? 4 : 0;
    G.m_sh = 6 * G.l;
    return G;
}

// ---- per-lane register image of one record ---------------------------------------------------------------------------------
template <int TYPE> struct Rec9;
template <> struct Rec9<GT_Q4_K> { u32x4 qs, hdr; };
template <> struct Rec9<GT_Q5_K> { u32x4 qs, hdr; uint32_t qh; };
template <> struct Rec9<GT_Q6_K> { u32x4 ql; u32x2 qh, sc; uint32_t d; };
template <> struct Rec9<GT_Q8_0> { u32x4 qs; u32x2 d; };
template <> struct Rec9<GT_Q4_0> { u32x4 qs; u32x2 d; };
template <int TYPE> DEV constexpr bool is_b32() { return TYPE == GT_Q8_0 || TYPE == GT_Q4_0; }
template <int TYPE> DEV constexpr uint32_t rec9_bytes() {
    return TYPE == GT_Q4_K ? 1152u : (TYPE == GT_Q5_K ? 1408u : (TYPE == GT_Q6_K ? 1680u : (TYPE == GT_Q8_0 ? 1088u : 576u)));
}

template <int TYPE> DEV Rec9<TYPE> rec9_load(const uint8_t* rec, const Lane9& G);
template <> DEV Rec9<GT_Q4_K> rec9_load<GT_Q4_K>(const uint8_t* rec, const Lane9& G) {
    Rec9<GT_Q4_K> R;
    R.hdr = ld_stream16(rec + G.o_hdr);
    R.qs = ld_stream16(rec + G.o16);
    return R;
}
template <> DEV Rec9<GT_Q5_K> rec9_load<GT_Q5_K>(const uint8_t* rec, const Lane9& G) {
    Rec9<GT_Q5_K> R;
    R.hdr = ld_stream16(rec + G.o_hdr5);
    R.qh = ld_stream4(rec + G.o4);
    R.qs = ld_stream16(rec + G.o16);
    return R;
}
template <> DEV Rec9<GT_Q6_K> rec9_load<GT_Q6_K>(const uint8_t* rec, const Lane9& G) {
    Rec9<GT_Q6_K> R;
    R.d = *(const uint16_t*)(rec + G.o_d6);
    R.sc = ld_stream8(rec + G.o_sc6);
    R.qh = ld_stream8(rec + G.o8);
    R.ql = ld_stream16(rec + G.o16);
    return R;
}

template <> DEV Rec9<GT_Q8_0> rec9_load<GT_Q8_0>(const uint8_t* rec, const Lane9& G) {
    Rec9<GT_Q8_0> R;
    R.d = ld_stream8(rec + G.o_d80);
    R.qs = ld_stream16(rec + G.o16);
    return R;
}
template <> DEV Rec9<GT_Q4_0> rec9_load<GT_Q4_0>(const uint8_t* rec, const Lane9& G) {
    Rec9<GT_Q4_0> R;
    R.d = ld_stream8(rec + G.o_d40);
    R.qs = ld_stream16(rec + G.o_qs40);
    return R;
}

// The scale / min words of a Q4_K / Q5_K block slot (quant.h LAYOUT_L9; engine_load.h:place_kblock9): scales 0..4 in W1, 5 and 6 in W3,
// scale 7 split over the spare bits; the eight mins are 48 contiguous bits of the pair W3:W2.  Two mad chains of four (hipcc would
// re-associate mul24 + add into v_mul x 2 + v_add3: eleven instructions of the 3.3-cycle class for the eight products instead of eight).
DEV int scaled_sum45(uint32_t W1, uint32_t W3, const int (&d)[8]) {
    int s0 = mul24((int)(W1 & 63u), d[0]);
    int s1 = mul24((int)bfe32(W1, 24, 6), d[4]);
    s0 = mad24((int)bfe32(W1, 6, 6), d[1], s0);
    s1 = mad24((int)bfe32(W3, 16, 6), d[5], s1);
    s0 = mad24((int)bfe32(W1, 12, 6), d[2], s0);
    s1 = mad24((int)bfe32(W3, 22, 6), d[6], s1);
    s0 = mad24((int)bfe32(W1, 18, 6), d[3], s0);
    const uint32_t sc7 = (W1 >> 30) | ((W3 >> 28) << 2);
    s1 = mad24((int)sc7, d[7], s1);
    return s0 + s1;
}
// min m_l of lane l: bits 6 l .. 6 l + 5 of W3:W2 (G.m_sh = 6 l)
DEV int min45(uint32_t W2, uint32_t W3, const Lane9& G) { return (int)(lshr64_lo(W3, W2, G.m_sh) & 63u); }
// Q6_K: sum over the eight vectors of the signed scale byte v (of the two words lo | hi) times d[v]; the byte selects fold into
// the multiplies (v_mul_i32_i24_sdwa with sext)
DEV int scaled_sum6(uint32_t lo, uint32_t hi, const int (&d)[8]) {
    int sum = 0;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const uint32_t w = v < 4 ? lo : hi;
        sum += mul24((int)(int8_t)((w >> (8 * (v & 3))) & 0xFFu), d[v]);
    }
    return sum;
}

// Integer work of this lane's block (4 * s + c of its row) -> the operands of the reference's f32 chain steps for that block:
// sv = (float)sumi[l], dv = y.d * fp16(d), and for the min term pv = (float)prod (Q4_K: prod[l >> 1]; Q5_K: the block total),
// mv = -y.d * fp16(dmin).  `img`: the block's image in LDS (Img9).
template <int TYPE>
DEV void step9(const Rec9<TYPE>& R, const int* img, const Lane9& G, float& sv, float& dv, float& mv, float& pv) {
    const u32x4 a_lo = *(const u32x4*)(img + G.a_w);        // vectors 0..3, elements 4l .. 4l+3
    const u32x4 a_hi = *(const u32x4*)(img + 32 + G.a_w);   // vectors 4..7
    const float yd = bits_to_f32((uint32_t)img[72]);
    int a8[8], w8[8], d8[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a8[k] = (int)a_lo[k]; a8[4 + k] = (int)a_hi[k]; }
    if constexpr (TYPE == GT_Q4_K || TYPE == GT_Q5_K) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t lo = R.qs[j] & 0x0F0F0F0Fu;
            uint32_t hi = (R.qs[j] >> 4) & 0x0F0F0F0Fu;
            if constexpr (TYPE == GT_Q5_K) {   // fifth bit: byte e, bit v of the lane's qh word = element 4l + e of vector v
                const int v0 = 2 * j, v1 = 2 * j + 1;
                lo |= (v0 < 4 ? (R.qh << (4 - v0)) : (R.qh >> (v0 - 4))) & 0x10101010u;
                hi |= (v1 < 4 ? (R.qh << (4 - v1)) : (R.qh >> (v1 - 4))) & 0x10101010u;
            }
            w8[2 * j] = (int)lo; w8[2 * j + 1] = (int)hi;
        }
        dot4x8(d8, w8, a8);
        const uint32_t W1 = R.hdr[1], W2 = R.hdr[2], W3 = R.hdr[3];
        sv = (float)scaled_sum45(W1, W3, d8);
        int p = mul24(min45(W2, W3, G), img[64 + G.l]);
        // prod[t] = m[2t] * q8s[2t] + m[2t+1] * q8s[2t+1] in the lanes of EVEN l (one DPP add from lane + 4: l + 1); the odd-l lanes keep a
        // value nobody reads — their copy of chain t never reaches the row result (the end-of-unit sums below read even l only)
        p = lane_up4_add(p);
        if constexpr (TYPE == GT_Q5_K) { p += lane_xor8(p); p += lane_xor16(p); }   // the block total (in the lanes of l = 0, 4 | 0: see v9_run)
        pv = (float)p;
        dv = mix_mul_f16lo(R.hdr[0], yd);        // y.d * fp16(d): conversion and product in one instruction, one rounding
        mv = mix_mulneg_f16hi(R.hdr[0], yd);     // -y.d * fp16(dmin)
    } else {
        // Q6_K: half n of the block = ql bytes 64n .. 64n+63 (ql[2n]: bytes 4l.. of the first 32, ql[2n+1]: of the second 32) and
        // qh bytes 32n + 4l ..; vector 4n + g: g&1 picks the ql word, g&2 its nibble, bits 2g of qh (reference k_quants.c:3800-3872)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const uint32_t A1 = R.ql[2 * n], A2 = R.ql[2 * n + 1], H = R.qh[n];
            w8[4 * n] = (int)((A1 & 0x0F0F0F0Fu) | ((H << 4) & 0x30303030u));
            w8[4 * n + 1] = (int)((A2 & 0x0F0F0F0Fu) | ((H << 2) & 0x30303030u));
            w8[4 * n + 2] = (int)(((A1 >> 4) & 0x0F0F0F0Fu) | (H & 0x30303030u));
            w8[4 * n + 3] = (int)(((A2 >> 4) & 0x0F0F0F0Fu) | ((H >> 2) & 0x30303030u));
        }
        // (q6 - 32) . a = q6 . a - 32 * (a0 + a1 + a2 + a3): the second term comes with the image
        const u32x4 b_lo = *(const u32x4*)(img + 80 + G.a_w);
        const u32x4 b_hi = *(const u32x4*)(img + 112 + G.a_w);
        int c8[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { c8[k] = (int)b_lo[k]; c8[4 + k] = (int)b_hi[k]; }
        dot4x8_add(d8, w8, a8, c8);
        // the lane's eight scales: scales[2v + (l >> 2)], v = 0..7 (the record keeps even and odd scales apart)
        sv = (float)scaled_sum6(R.sc[0], R.sc[1], d8);
        dv = yd * f16_bits_to_f32((uint16_t)(R.d & 0xFFFF));
        mv = 0.0f;
        pv = 0.0f;
    }
}

// 32-element block types: one record = 16 blocks of the lane's row = four chain sub-steps; sub-step t: the quad's lanes c = 0..3 hold
// blocks 4t + c.  Reference ggml.c:3321 (Q8_0) / :2428 (Q4_0), AVX2: acc[l] = fma(fp16(x.d) * fp16(y.d), (float)sumi[l], acc[l]) block
// after block, sumi[l] = the four products of elements 4l .. 4l+3 (Q4_0: (nibble - 8) . a).
template <int TYPE>
DEV float step9b(const Rec9<TYPE>& R, const int* img, const Lane9& G, float acc) {
    const u32x4 a4 = *(const u32x4*)(img + G.a_b);
    const f32x4 yd = *(const f32x4*)(img + 256 + 4 * G.c);
    int w4[4], av[4], d4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) av[t] = (int)a4[t];
    if constexpr (TYPE == GT_Q8_0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) w4[t] = (int)R.qs[t];
        dot4x4(d4, w4, av);
    } else {
        const u32x4 b4 = *(const u32x4*)(img + 128 + G.a_b);
        int bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { w4[t] = (int)((R.qs[t] >> G.sh40) & 0x0F0F0F0Fu); bv[t] = (int)b4[t]; }
        dot4x4_add(d4, w4, av, bv);
    }
    const uint32_t dw[4] = {R.d[0] & 0xFFFFu, R.d[0] >> 16, R.d[1] & 0xFFFFu, R.d[1] >> 16};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = quad_chain4(acc, f16_bits_to_f32((uint16_t)dw[t]) * yd[t], (float)d4[t]);
    return acc;
}

// Prologue of the 32-block types: (RMSNorm | LayerNorm | nothing) -> Q8_0 images.  Reference ggml.c:1208-1300 (AVX2 path): per 32
// elements d = amax / 127 stored as fp16, id = 127 / amax (0 when amax == 0), q = round-half-even(x * id).  8 lanes per block (lane l
// holds elements 4l .. 4l+3), 128 blocks per pass of the 1024 threads.
template <int MAXK> struct Pro9b {
    static constexpr int PASSES = MAXK / 32 / 128;
    static constexpr bool EARLY_W = MAXK <= 16384;
    float4 x[PASSES];
    float4 w[EARLY_W ? PASSES : 1];
};
template <int MAXK>
DEV void pro9b_load(Pro9b<MAXK>& P, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro) {
    const int tid = (int)threadIdx.x, l = tid & 7, nblk = K >> 5;
#pragma unroll
    for (int ps = 0; ps < Pro9b<MAXK>::PASSES; ++ps) {
        const int b = (tid >> 3) + ps * 128;
        P.x[ps] = float4{0.f, 0.f, 0.f, 0.f};
        if constexpr (Pro9b<MAXK>::EARLY_W) P.w[ps] = float4{0.f, 0.f, 0.f, 0.f};
        if (b < nblk) {
            P.x[ps] = *(const float4*)(x + b * 32 + l * 4);
            if constexpr (Pro9b<MAXK>::EARLY_W) {
                if (pro != PRO_PLAIN) P.w[ps] = *(const float4*)(nw + b * 32 + l * 4);
            }
        }
    }
}
template <int MAXK, bool Q4BIAS, bool EMB>
DEV void pro9b_finish(Img9<MAXK>& L, Pro9b<MAXK>& P, const float* __restrict__ nw, const float* __restrict__ nbias, int K, int pro, float eps,
                      float* __restrict__ emb_out) {
    constexpr int PASSES = Pro9b<MAXK>::PASSES;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, l = tid & 7;
    const int nblk = K >> 5, nimg = ((nblk + 15) >> 4) << 4;
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {   // ggml.c:10700-10716: double sum, f32 mean, 1/sqrtf
        double s = 0.0;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if ((tid >> 3) + ps * 128 < nblk) {
                const float4 q = P.x[ps];
                s += (double)(q.x * q.x); s += (double)(q.y * q.y); s += (double)(q.z * q.z); s += (double)(q.w * q.w);
            }
        }
        s = wave_sum_fast(s);
        if (lane == 0) L.red[0][wv] = s;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += L.red[0][w];
        const float mean = (float)(tot / (double)K);
        scale = 1.0f / sqrtf(mean + eps);
    } else if (pro == PRO_LAYERNORM) {   // ggml.c:10605-10654
        double s1 = 0.0;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
            if ((tid >> 3) + ps * 128 < nblk) { const float4 q = P.x[ps]; s1 += (double)q.x; s1 += (double)q.y; s1 += (double)q.z; s1 += (double)q.w; }
        s1 = wave_sum_fast(s1);
        if (lane == 0) L.red[0][wv] = s1;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += L.red[0][w];
        const float mean = (float)(tot / (double)K);
        double s2 = 0.0;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if ((tid >> 3) + ps * 128 < nblk) {
                float4& q = P.x[ps];
                q.x -= mean; q.y -= mean; q.z -= mean; q.w -= mean;
                s2 += (double)(q.x * q.x); s2 += (double)(q.y * q.y); s2 += (double)(q.z * q.z); s2 += (double)(q.w * q.w);
            }
        }
        s2 = wave_sum_fast(s2);
        if (lane == 0) L.red[1][wv] = s2;
        __syncthreads();
        double tot2 = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot2 += L.red[1][w];
        const float variance = (float)(tot2 / (double)K);
        scale = 1.0f / sqrtf(variance + eps);
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int b = (tid >> 3) + ps * 128;
        const bool live = b < nblk;
        float4 t = live ? P.x[ps] : float4{0.f, 0.f, 0.f, 0.f};
        if (live && pro != PRO_PLAIN) {
            float4 w4;
            if constexpr (Pro9b<MAXK>::EARLY_W) w4 = P.w[ps];
            else w4 = *(const float4*)(nw + b * 32 + l * 4);
            t.x = (t.x * scale) * w4.x; t.y = (t.y * scale) * w4.y; t.z = (t.z * scale) * w4.z; t.w = (t.w * scale) * w4.w;
            if (pro == PRO_LAYERNORM) {
                const float4 b4 = *(const float4*)(nbias + b * 32 + l * 4);
                t.x += b4.x; t.y += b4.y; t.z += b4.z; t.w += b4.w;
            }
            if constexpr (EMB) {
                if (emb_out && blockIdx.x == 0) *(float4*)(emb_out + b * 32 + l * 4) = t;
            }
        }
        float amax = fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w)));
        amax = fmaxf(amax, lane_xor1(amax));
        amax = fmaxf(amax, lane_xor2(amax));
        amax = fmaxf(amax, lane_xor4(amax));
        const float d = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        const int q0 = (int)__builtin_rintf(t.x * id), q1 = (int)__builtin_rintf(t.y * id);
        const int q2 = (int)__builtin_rintf(t.z * id), q3 = (int)__builtin_rintf(t.w * id);
        if (b < nimg) {   // blocks past the row's end (the padding of its last record): zero quants, y.d = 0
            const int packed = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
            const int i16 = b & 15, tt = i16 >> 2, c = i16 & 3;
            int* img = &L.blk[(b >> 4) * kImg9bStride];
            img[(c * 8 + l) * 4 + tt] = packed;
            if constexpr (Q4BIAS) img[128 + (c * 8 + l) * 4 + tt] = sdot4(packed, (int)0xF8F8F8F8u, 0);   // -8 * (sum of the four quants)
            if (l == 0) img[256 + c * 4 + tt] = (int)f32_to_bits(f16_bits_to_f32(f32_to_f16_bits(d)));
        }
    }
    __syncthreads();
}

// ---- prologue -----------------------------------------------------------------------------------------------------------------
// 16 lanes per 256-block (lane `sub` owns the 16 consecutive activations 16 sub .. 16 sub + 15), a wave = 4 blocks, the workgroup =
// 64 blocks per round: K <= 16384 is ONE round.  What every workgroup of a launch recomputes is issue-bound, so it is written for
// instruction count: the block-scalar work (largest / smallest value, the two divisions) is shared by four blocks per wave, packed
// f32 multiplies, the quant byte is the low byte of the magic-number sum (clamped as a float), sums of 16 by dot4.  Measured in
// round 4 (profiles/r04_decode_ablations.txt): ending the prologue 400 cycles earlier moves no launch time — it runs under the
// latency of the launch's first weight records; what it must not do is delay the REQUESTS (pro9_load, the barrier in the kernel).
template <int MAXK, bool EW, int NW = 16> struct Pro9 {
    static constexpr int ROUNDS = MAXK / 256 / (4 * NW);   // NW waves x 4 blocks per round
    // the norm weights travel with the activations (requested later they would queue behind the weight stream: 3000 cycles) —
    // except in the widest instantiation (K > 16384: ffn_down of the 70B class, no norm in front of it)
    static constexpr bool EARLY_W = MAXK <= 16384;
    float4 x[ROUNDS][4];
    float4 w[EARLY_W ? ROUNDS : 1][4];
};

// Part 1: request this thread's activations (and norm weights) — nothing waits here.  The first instructions of the kernel:
// whatever is requested later queues behind the weight stream.  Wave-uniform branches only; a lane past the last block reads the
// last block again (its values are replaced by zeros: the padding slots of a row's last record).
template <int MAXK, bool EW, int NW>
DEV void pro9_load(Pro9<MAXK, EW, NW>& P, const float* __restrict__ x, const float* __restrict__ nw, int K, int pro, int wv, int lane) {
    const int nblk = K >> 8, nimg = ((nblk + 3) >> 2) << 2;
    constexpr int RB = 4 * NW;   // blocks per round
#pragma unroll
    for (int rd = 0; rd < Pro9<MAXK, EW, NW>::ROUNDS; ++rd) {
        if (RB * rd + 4 * wv < nimg) {
            int b = RB * rd + 4 * wv + (lane >> 4);
            b = b < nblk ? b : nblk - 1;
            const float* src = x + b * 256 + (lane & 15) * 16;
#pragma unroll
            for (int k = 0; k < 4; ++k) P.x[rd][k] = *(const float4*)(src + 4 * k);
            if constexpr (Pro9<MAXK, EW, NW>::EARLY_W) {
                if (pro != PRO_PLAIN) {
                    const float* wsrc = nw + b * 256 + (lane & 15) * 16;
#pragma unroll
                    for (int k = 0; k < 4; ++k) P.w[rd][k] = *(const float4*)(wsrc + 4 * k);
                }
            }
        }
    }
}

DEV double row16_sum(double v) { v += lane_xor1(v); v += lane_xor2(v); v += lane_xor4(v); v += lane_xor8(v); return v; }

// The norm's sum over the whole vector from per-lane partials `sr[rd]` of block RB * rd + 4 * wv + (lane >> 4) (zero for a block past
// the row's end).  Every 16-lane row reduces ITS block (four DPP steps, no cross-row traffic), one lane per row parks the block's sum,
// the waves that own blocks meet at the arrival counter, then lane `sub` of every row takes the blocks sub, sub + 16, ... and the row
// reduces again: every lane holds the total.  (Round 3's form — a 64-lane reduce through two ds_bpermute rounds, one slot per wave,
// a serial loop of dependent LDS reads over the live waves — took 1600 cycles from "activations arrived" to "scale known".)  The
// grouping differs from the reference's sequential ggml_float sum: the double sum of float squares / values is exact — hence
// order-free — only while the addends of a row span less than ~2^26 in magnitude (53-bit significand against 48-bit squares of 4096+
// addends); beyond that a regrouping can change the last bit of the double, which survives the rounding to float with probability
// ~1e-9 per op (the one place where the reference's order is not reproduced).
template <int MAXK, int NW, int ROUNDS>
DEV double pro9_total(double* red, unsigned* cnt, unsigned target, const double (&sr)[ROUNDS], int nimg, int wv, int lane) {
    constexpr int RB = 4 * NW;
    const int sub = lane & 15;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        if (RB * rd + 4 * wv < nimg) {
            const double r = row16_sum(sr[rd]);
            if (sub == 0) red[RB * rd + 4 * wv + (lane >> 4)] = r;
        }
    }
    lds_signal(cnt, lane, 1u);
    lds_wait_ge(cnt, target);
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < MAXK / 256 / 16; ++j) {
        const int b = sub + 16 * j;
        const double v = red[b < nimg ? b : 0];
        t += b < nimg ? v : 0.0;
    }
    return row16_sum(t);
}

// Part 2: (RMSNorm | LayerNorm | nothing) -> Q8_K image in LDS.  Reference k_quants.c:1191-1226 with the build's fused
// `iscale * x + 12582912.f`, RMSNorm ggml.c:10700-10716, LayerNorm ggml.c:10605-10654 (arithmetic of generation 7's prologue; the
// double-precision sums are regrouped: see pro9_total).  Ends with a workgroup barrier.  `emb_out` (workgroup 0 only): the
// normalised vector as f32 — the final-norm "embeddings" output of the ABI, produced by the lm_head launch (EMB instantiation).
template <int MAXK, bool LN, bool EMB, bool EW, bool Q6IMG, int NW>
DEV void pro9_finish(Img9<MAXK>& L, Pro9<MAXK, EW, NW>& P, const float* __restrict__ nw, const float* __restrict__ nbias, int K, int pro,
                     float eps, float* __restrict__ emb_out, int wv, int lane, bool trace, unsigned long long (&ts)[4]) {
    constexpr int ROUNDS = Pro9<MAXK, EW, NW>::ROUNDS;
    constexpr int RB = 4 * NW;   // blocks per round
    const int nblk = K >> 8, nimg = ((nblk + 3) >> 2) << 2;
    const int sub = lane & 15;
    const bool wave_live = 4 * wv < nimg;   // this wave owns blocks in round 0 (a wave live in round 1 is live in round 0)
    float scale = 1.0f;
    // The norm's sums: only the waves that own blocks meet (LDS arrival counter, cleared before the kernel's first barrier); a wave
    // without blocks goes straight to the final barrier — it may still be busy issuing its first weight requests.
    const unsigned n_live = (unsigned)(nimg >> 2 < NW ? nimg >> 2 : NW);
    const bool pow2 = (K & (K - 1)) == 0;
    const double inv_k = 1.0 / (double)K;   // exact for a power of two: tot * inv_k == tot / K bit for bit
    if (pro == PRO_RMSNORM && wave_live) {
        double sq[ROUNDS];
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            sq[rd] = 0.0;
            if (RB * rd + 4 * wv < nimg) {
                double sr = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 q = P.x[rd][k];
                    sr += (double)(q.x * q.x); sr += (double)(q.y * q.y); sr += (double)(q.z * q.z); sr += (double)(q.w * q.w);
                }
                sq[rd] = (RB * rd + 4 * wv + (lane >> 4) < nblk) ? sr : 0.0;
            }
        }
        V9_STAMP(trace, ts[0]);   // the activations have arrived (their squares are summed)
        const double tot = pro9_total<MAXK, NW, ROUNDS>(L.red[0], &L.cnt, n_live, sq, nimg, wv, lane);
        const float mean = (float)(pow2 ? tot * inv_k : tot / (double)K);
        scale = 1.0f / sqrtf(mean + eps);
        V9_STAMP(trace, ts[1]);   // the norm's scale is known
    }
    if constexpr (LN) {
        if (pro == PRO_LAYERNORM && wave_live) {
            double m1[ROUNDS];
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                m1[rd] = 0.0;
                if (RB * rd + 4 * wv < nimg) {
                    double sr = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 q = P.x[rd][k];
                        sr += (double)q.x; sr += (double)q.y; sr += (double)q.z; sr += (double)q.w;
                    }
                    m1[rd] = (RB * rd + 4 * wv + (lane >> 4) < nblk) ? sr : 0.0;
                }
            }
            const double tot = pro9_total<MAXK, NW, ROUNDS>(L.red[0], &L.cnt, n_live, m1, nimg, wv, lane);
            const float mean = (float)(pow2 ? tot * inv_k : tot / (double)K);
            double m2[ROUNDS];
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                m2[rd] = 0.0;
                if (RB * rd + 4 * wv < nimg) {
                    double sr = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float4& q = P.x[rd][k];
                        q.x -= mean; q.y -= mean; q.z -= mean; q.w -= mean;
                        sr += (double)(q.x * q.x); sr += (double)(q.y * q.y); sr += (double)(q.z * q.z); sr += (double)(q.w * q.w);
                    }
                    m2[rd] = (RB * rd + 4 * wv + (lane >> 4) < nblk) ? sr : 0.0;
                }
            }
            const double tot2 = pro9_total<MAXK, NW, ROUNDS>(L.red[1], &L.cnt, 2u * n_live, m2, nimg, wv, lane);
            const float variance = (float)(pow2 ? tot2 * inv_k : tot2 / (double)K);
            scale = 1.0f / sqrtf(variance + eps);
        }
    }
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        if (RB * rd + 4 * wv < nimg) {
            const int b = RB * rd + 4 * wv + (lane >> 4);
            const bool live = b < nblk;            // row-uniform; a dead row of a live wave writes a zero image (b < nimg) or nothing
            const int bc = live ? b : nblk - 1;
            float t[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 q = P.x[rd][k];
                if (pro != PRO_PLAIN) {
                    float4 w4;
                    if constexpr (Pro9<MAXK, EW, NW>::EARLY_W) w4 = P.w[rd][k];
                    else w4 = *(const float4*)(nw + bc * 256 + sub * 16 + k * 4);
                    q.x = (q.x * scale) * w4.x; q.y = (q.y * scale) * w4.y; q.z = (q.z * scale) * w4.z; q.w = (q.w * scale) * w4.w;
                    if constexpr (LN) {
                        if (pro == PRO_LAYERNORM) {
                            const float4 b4 = *(const float4*)(nbias + bc * 256 + sub * 16 + k * 4);
                            q.x += b4.x; q.y += b4.y; q.z += b4.z; q.w += b4.w;
                        }
                    }
                    if constexpr (EMB) {   // a store here is a pending write at the entry of the streaming loop: lm_head instantiation only
                        if (emb_out && blockIdx.x == 0 && live) *(float4*)(emb_out + b * 256 + sub * 16 + k * 4) = q;
                    }
                }
                t[4 * k] = live ? q.x : 0.0f; t[4 * k + 1] = live ? q.y : 0.0f; t[4 * k + 2] = live ? q.z : 0.0f; t[4 * k + 3] = live ? q.w : 0.0f;
            }
            // amax and the element that attains it first (k_quants.c:1198-1204: `if (ax > amax) { amax = ax; max = x[j]; }` — the FIRST
            // element of largest magnitude keeps its sign).  Largest and smallest SIGNED value of the block instead of the largest
            // magnitude + a search for its first position: amax = max(hi, -lo), and unless hi == -lo (both +amax and -amax occur: then the
            // order decides, the wave takes the search below — in practice never) the sign follows from which of the two it is.
            float hi = fmaxf(fmaxf(t[0], t[1]), t[2]), lo = fminf(fminf(t[0], t[1]), t[2]);
#pragma unroll
            for (int e = 3; e < 15; e += 2) { hi = fmaxf(fmaxf(hi, t[e]), t[e + 1]); lo = fminf(fminf(lo, t[e]), t[e + 1]); }
            hi = fmaxf(hi, t[15]); lo = fminf(lo, t[15]);
#ifdef V9_PROTRACE
            if (trace && rd == 0 && pro == PRO_PLAIN) { reg_fence(hi, t[0], t[1], t[2]); ts[0] = clock64_dev(); }   // the activations have arrived
#endif
            hi = fmaxf(hi, lane_xor1(hi)); lo = fminf(lo, lane_xor1(lo));
            hi = fmaxf(hi, lane_xor2(hi)); lo = fminf(lo, lane_xor2(lo));
            hi = fmaxf(hi, lane_xor4(hi)); lo = fminf(lo, lane_xor4(lo));
            hi = fmaxf(hi, lane_xor8(hi)); lo = fminf(lo, lane_xor8(lo));
            const float amax = fmaxf(hi, -lo);
            float maxv = hi == amax ? hi : lo;
            if (__ballot(hi == -lo && amax != 0.0f) != 0ull) {   // a block holds +amax and -amax: the first of them, by position
                float am = fabsf(t[0]);
#pragma unroll
                for (int e = 1; e < 16; ++e) am = fmaxf(am, fabsf(t[e]));
                const unsigned long long hit = __ballot(am == amax);
                const unsigned row_bits = (unsigned)((hit >> (lane & 48)) & 0xFFFFu);
                const int first = __ffsll((unsigned long long)row_bits) - 1;
                float mine = 0.0f;
#pragma unroll
                for (int e = 15; e >= 0; --e) mine = (fabsf(t[e]) == amax) ? t[e] : mine;
                uint32_t mb = sub == first ? f32_to_bits(mine) : 0u;
                mb |= lane_xor1(mb); mb |= lane_xor2(mb); mb |= lane_xor4(mb); mb |= lane_xor8(mb);
                maxv = bits_to_f32(mb);
            }
            const bool nz = amax != 0.0f;
            const float iscale = nz ? -128.f / maxv : 0.0f;
            const float d = nz ? 1.0f / iscale : 0.0f;
            // nearest_int(iscale * x) = bits(fma(iscale, x, 1.5 * 2^23)) - bits(1.5 * 2^23): the sum stays in the binade of ulp 1, so the
            // quant is its low byte and MIN(127, .) is a float minimum against 1.5 * 2^23 + 127
            uint32_t packed[4];
            int s16 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t by[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) by[e] = f32_to_bits(fminf(fmaf(iscale, t[4 * k + e], 12582912.f), 12583039.f));
                packed[k] = pack_low_bytes(by[0], by[1], by[2], by[3]);
                s16 = sdot4((int)packed[k], 0x01010101, s16);
            }
            int neg32[4] = {0, 0, 0, 0};   // Q6_K launches: -32 * (sum of a word's four quants), the start value of its dot product
            if constexpr (Q6IMG) {
#pragma unroll
                for (int k = 0; k < 4; ++k) neg32[k] = sdot4((int)packed[k], (int)0xE0E0E0E0u, 0);
            }
            const int s32 = s16 + lane_xor1(s16);
            V9_STAMP(trace && rd == 0, ts[2]);   // round 0 quantized (in registers)
            if (b < nimg) {
                const int v = sub >> 1, l0 = 4 * (sub & 1);
                int* img = &L.blk[b * kImg9Stride];
                int* qd = img + 32 * (v >> 2) + 4 * l0 + (v & 3);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    qd[4 * k] = (int)packed[k];
                    if constexpr (Q6IMG) qd[80 + 4 * k] = neg32[k];
                }
                if ((sub & 1) == 0) img[64 + v] = s32;
                if (sub == 0) img[72] = (int)f32_to_bits(d);
            }
        }
    }
    V9_STAMP(trace, ts[3]);   // images written, at the barrier
    __syncthreads();
}

template <bool B> struct V9Req { static constexpr bool value = B; };

// Ring slots per wave (records in flight) of the three launch families (engine.cc:launch_matvec_kq has the measurements): single-type
// K-quant launches, the two-type (QKV) launch, the 32-block types.
#ifndef V9_NS_K
#define V9_NS_K 3
#endif
#ifndef V9_NS_K2
#define V9_NS_K2 4
#endif
#ifndef V9_NS_B
#define V9_NS_B 4
#endif

// The units of one wave, as a list of items of the launch's concatenated unit list.
// Items9: first, first + stride, ... < end — the plain launch form (a workgroup's waves interleave with every other workgroup's).
struct Items9 {
    int first, stride, end;
    DEV int count() const { return first < end ? (end - first + stride - 1) / stride : 0; }
    DEV int at(int k) const { return first + k * stride; }
};
// All units of one wave (`items`).  base / g0: first record and first item of the wave's type group.
// XQ (fused QKV + attention launch): the RoPE / V epilogues also publish their fp16 results as tagged granules (a.xq: kernels_qa9.h).
template <int TYPE, int MAXK, bool TWO, int NS, bool PICK, bool XQ = false, class Items, class Pro>
DEV void v9_run(const MatvecArgs& a, SmemV9<MAXK>& SM, const uint8_t* base, int g0, Items& items, int lane, int wv,
                Pro pro) {
    constexpr bool B32 = is_b32<TYPE>();
    constexpr bool mins = TYPE == GT_Q4_K || TYPE == GT_Q5_K;
    constexpr uint32_t REC = rec9_bytes<TYPE>();
    const int spu = B32 ? ((a.K >> 5) + 15) >> 4 : ((a.K >> 8) + 3) >> 2;
    const size_t unit_bytes = (size_t)spu * REC;
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv;
    const int nu = items.count();   // units of this wave (host: <= kV9MaxUnits)
    // ---- prefetch cursor: the step whose record is requested next (stays on the last record once everything is requested) ----
    int pf_k = 0, pf_s = 0, pf_left = nu * spu;
    const uint8_t* pf_ptr = base + (size_t)((nu > 0 ? items.at(0) : g0) - g0) * unit_bytes;
    Rec9<TYPE> ring[NS];
    auto issue = [&](Rec9<TYPE>& slot, const Lane9& GG) __attribute__((always_inline)) {
        slot = rec9_load<TYPE>(pf_ptr, GG);   // unconditional, same instruction count every step: a conditional load in the loop makes hipcc wait vmcnt(0) per step
        if (pf_left > 1) {
            --pf_left;
            pf_ptr += REC;
            if (++pf_s == spu) {
                pf_s = 0;
                ++pf_k;
                pf_ptr = base + (size_t)(items.at(pf_k) - g0) * unit_bytes;
            }
        }
    };
    // Records requested before the prologue: about 50 KB per CU over the 16 waves — what the CU's memory pipeline takes without
    // blocking the waves in the issue of their requests (four per wave: the last wave reaches the prologue thousands of cycles late).
    // A wave without units requests nothing (wave-uniform branch; it only takes part in the prologue).
#ifndef V9_PRE
#define V9_PRE ((TYPE == GT_Q6_K || B32) ? 2 : 3)   // measured on the 7B: 1 -> 687, 2 -> 715, 3 -> 737, 4 -> 730 tok/s
#endif
    constexpr int PRE = (V9_PRE) < NS ? (V9_PRE) : NS;
    if (nu == 0) {   // its own copy of the prologue: the path with requests below stays free of conditional loads
        unsigned long long ts0[4];
        pro(false, ts0);
        if constexpr (PICK) { if (a.pick_ws && lane == 0) ((unsigned long long*)a.pick_ws)[(int)blockIdx.x * 16 + wv] = 0ull; }   // no rows: no candidate
        return;
    }
    {
        const Lane9 G0 = lane9(lane);
#pragma unroll
        for (int k = 0; k < PRE; ++k) issue(ring[k], G0);
    }
    // trace stamps are kept in registers until the loop is over (a store before it would be a pending write at its entry)
    const unsigned long long t1 = trace ? clock64_dev() : 0ull;
    unsigned long long ts[4] = {0ull, 0ull, 0ull, 0ull};
    pro(trace, ts);
    const unsigned long long t2 = trace ? clock64_dev() : 0ull;
    // the lane constants are recomputed behind the prologue instead of living through it (a dozen registers the two-type
    // instantiations do not have: they spilled)
    const Lane9 G = lane9(opaque_int(lane));
    // (deep rings, NS > 4: the host launches them only where NS divides the records of a unit — every request below and in the loop
    // names a record that exists; a branch around a request would make hipcc wait vmcnt(0) before the first step: measured, 18 500
    // instead of 15 000 cycles for ffn_down)
#pragma unroll
    for (int k = PRE; k < NS; ++k) issue(ring[k], G);
    // ---- epilogue operands of lane l = (unit l >> 1, row l & 1), requested before the loop (its body must not contain a load
    //      besides the ring's); unconditional loads: operands a lane does not need are read from the activation vector ----
    const int p1 = a.njobs > 1 ? a.job[1].pair0 : 0x7fffffff, p2 = a.njobs > 2 ? a.job[2].pair0 : 0x7fffffff;
    const int e_it = items.at(lane >> 1);
    const bool e_valid = (lane >> 1) < nu;
    const int e_j = e_it >= p2 ? 2 : (e_it >= p1 ? 1 : 0);
    const int e_p0 = e_j == 2 ? p2 : (e_j == 1 ? p1 : 0);
    // the jobs' fields as scalars first: selected straight out of the argument array, hipcc turns the select into a per-lane LOAD
    // from the kernel-argument segment — a vector memory load younger than the ring's, i.e. an `s_waitcnt vmcnt(0)` on all four
    // records before the first step (generation 7 had it: profiles/r02 traces show its math starting when the whole ring had landed)
    const int jM0 = uniform_int(a.job[0].w.M), jM1 = uniform_int(a.job[1].w.M), jM2 = uniform_int(a.job[2].w.M);
    const int jE0 = uniform_int(a.job[0].epi), jE1 = uniform_int(a.job[1].epi), jE2 = uniform_int(a.job[2].epi);
    const int e_M = e_j == 2 ? jM2 : (e_j == 1 ? jM1 : jM0);
    const int e_epi = e_j == 2 ? jE2 : (e_j == 1 ? jE1 : jE0);
    const int e_u = e_it - e_p0;
    const int e_r = a.gateup ? e_u : 2 * e_u + (lane & 1);   // output row of this lane
    const bool e_own = e_valid && e_r < e_M;
    const bool need_res = e_own && (e_epi == EPI_ADD || e_epi == EPI_ADD2), need_res2 = e_own && e_epi == EPI_ADD2;
    const bool need_rope = e_own && (e_epi == EPI_ROPE_Q || e_epi == EPI_ROPE_K);
    const bool need_bias = e_own && (e_epi == EPI_BIAS_STORE || e_epi == EPI_BIAS_ADD || e_epi == EPI_BIAS_GELU);
    const bool need_res_b = e_own && e_epi == EPI_BIAS_ADD;
    // the cursor position: only launches that write the KV cache / rotate (QKV) read it; after the barriers, so that this
    // scalar-cache round trip holds up no other wave
    bool need_pos = false;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) need_pos = need_pos || (jj < a.njobs && (a.job[jj].epi == EPI_ROPE_Q || a.job[jj].epi == EPI_ROPE_K || a.job[jj].epi == EPI_V));
    int pos = (need_pos && a.pos && !XQ) ? sload_i32(a.pos) : 0;
    if constexpr (XQ) { items.load_scalars(a); pos = items.cursor_pos(); }   // the position and the token epoch, one scalar-cache round trip (kernels_qa9.h)
    const bool want_res = need_res || (B32 && need_res_b);
    const float e_res = (want_res ? a.res : a.x)[want_res ? e_r : 0];
    float e_bias = 0.0f;   // the bias epilogues exist for the legacy (gpt2 / starcoder / mpt) graphs only: 32-block weight types
    if constexpr (B32) e_bias = (need_bias ? a.bias : a.x)[need_bias ? e_r : 0];
    const float e_res2 = (need_res2 ? a.res2 : a.x)[need_res2 ? e_r : 0];
    const float2 e_cs = *(const float2*)((need_rope ? a.rope_cs : a.x) +
                                         (need_rope ? ((size_t)pos * (a.head_dim >> 1) + ((e_r % a.head_dim) >> 1)) * 2 : 0));
    // ---- consume: one flat sequence of nu * spu steps, in groups of four (the ring slots are compile-time) ----
    const int total = nu * spu, groups = (total + NS - 1) / NS;
    int s = 0, ui = 0;
    float acc = 0.0f, accm = 0.0f;
    const int* img0 = &SM.L.blk[G.c * kImg9Stride];
    // One step: block math of ring slot R (blocks 4s .. 4s+3 of the wave's current unit), the slot's next request (REQ), the
    // reference's chain steps of the four blocks in order (operands from the quad's lanes c = 0..3), and at the end of a unit the
    // reference's reduction tree (hsum_float_8: (x_l + x_{l+4}), then l ^ 2, then l ^ 1; l sits in lane bits 2..4) — the row results
    // wait in LDS for the epilogue pass.  A surplus step of the last group re-processes the last record; its result is dropped.
    auto step = [&](Rec9<TYPE>& R, auto REQ) __attribute__((always_inline)) {
        if constexpr (B32) {   // 16 blocks of the row: four chain sub-steps inside
            acc = step9b<TYPE>(R, &SM.L.blk[s * kImg9bStride], G, acc);
            if constexpr (decltype(REQ)::value) {
                float z0 = 0.f, z1 = 0.f, z2 = 0.f;
                reg_fence(acc, z0, z1, z2);
                issue(R, G);
            }
        } else {
            float sv, dv, mv, pv;
            step9<TYPE>(R, img0 + s * (4 * kImg9Stride), G, sv, dv, mv, pv);
            if constexpr (decltype(REQ)::value) {
                reg_fence(sv, dv, mv, pv);   // every use of the record is over before its registers are given to the next load
                issue(R, G);
            }
            {
            acc = quad_chain4(acc, dv, sv);
            if constexpr (mins) accm = quad_chain4(accm, mv, pv);
            }
        }
        if (s + 1 < spu) { ++s; return; }
        const float t4 = acc + lane_xor16(acc);
        const float t2 = t4 + lane_xor8(t4);
        float res = t2 + lane_xor4(t2);
        if constexpr (mins) {
            if constexpr (TYPE == GT_Q4_K) {   // acc_m[t] lives in lanes l = 2t, 2t+1: (acc_m0 + acc_m2) + (acc_m1 + acc_m3)
                const float wsum = accm + lane_xor16(accm);
                accm = wsum + lane_xor8(wsum);
            }
            res = res + accm;
        }
        if ((lane & 31) == 0 && ui < nu) SM.RES[wv][2 * ui + G.row] = res;
        ++ui;
        acc = 0.0f; accm = 0.0f;
        s = 0;
    };
    // All groups but the last: every step requests the record its slot holds four steps later — the loop contains no other vector
    // memory instruction and no conditional one, so hipcc counts the ring with `s_waitcnt vmcnt(N)`.  A request
    // past the wave's last record (at most three, and none when a unit is a multiple of four records) reads that record again.
    // The last group requests nothing: the memory pipeline of the CU belongs to the waves that still have records to fetch.
    for (int g = 0; g + 1 < groups; ++g) {
#pragma unroll
        for (int k = 0; k < NS; ++k) step(ring[k], V9Req<true>{});
    }
    {   // the last group: no requests, and no surplus steps (a branch around block math is harmless here: no load follows)
        const int left = total - (groups - 1) * NS;
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (k < left) step(ring[k], V9Req<false>{});
    }
    if (trace) { tr[1] = t1; tr[2] = t2; tr[3] = clock64_dev(); tr[8] = ts[0]; tr[9] = ts[1]; tr[10] = ts[2]; tr[11] = ts[3]; }
    // ---- epilogue pass: lane l finishes row (l & 1) of unit l >> 1 ----
    wave_lds_sync();
    const float res = SM.RES[wv][lane];
    const float other = lane_xor1(res);   // the unit's other row (RoPE partner / up projection)
    if constexpr (PICK) {   // head launch: this wave's first maximum of the logits it is about to store (v9_pick_store)
        if (a.pick_ws) {
            const float bv = (e_own && e_epi == EPI_STORE && res > -INFINITY) ? res : -INFINITY;
            v9_pick_store(a, wv, lane, bv, bv > -INFINITY ? e_r : 0x7fffffff);
        }
    }
    if (a.gateup) {   // fused matrix: row 0 of the pair = gate row u, row 1 = up row u
        if (e_own && (lane & 1) == 0) a.out[e_r] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(res)]) * other;
        return;
    }
    if (!e_own) return;
    if constexpr (XQ) {   // fused QKV + attention launch: the pair's two fp16 results as one tagged granule (kernels_qa9.h); a QKV unit is a whole
                          // row pair, so both lanes of a pair are here
        float val = res;
        if (e_epi == EPI_ROPE_Q || e_epi == EPI_ROPE_K) {
            if (a.rope_neox) val = (e_r & 1) ? fmaf(other, e_cs.y, res * e_cs.x) : fmaf(res, e_cs.x, -(other * e_cs.y));   // (the NEOX pair of a reordered row pair: below)
            else val = (e_r & 1) ? fmaf(res, e_cs.x, other * e_cs.y) : fmaf(res, e_cs.x, -(other * e_cs.y));
        }
        const uint32_t hb = f32_to_f16_bits(val), ho = lane_xor1(hb);
        if ((lane & 1) == 0) items.publish(lane >> 1, hb | (ho << 16));
    }
    if (e_epi == EPI_ADD) {
        a.out[e_r] = res + e_res;
    } else if (e_epi == EPI_STORE) {
        a.out[e_r] = res;
    } else if (e_epi == EPI_V) {
        a.vcache[(size_t)e_r * a.v_stride + pos] = f32_to_f16_bits(res);
    } else if (e_epi == EPI_GELU) {
        a.out[e_r] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(res)]);
    } else if (e_epi == EPI_ADD2) {
        a.out[e_r] = (res + e_res) + e_res2;
    } else if (e_epi == EPI_BIAS_STORE) {
        a.out[e_r] = e_bias + res;
    } else if (e_epi == EPI_BIAS_ADD) {
        a.out[e_r] = (e_bias + res) + e_res;
    } else if (e_epi == EPI_BIAS_GELU) {
        a.out[e_r] = f16_bits_to_f32(a.gelu_tab[f32_to_f16_bits(e_bias + res)]);
    } else if (a.rope_neox) {   // falcon: rows (2i, 2i + 1) hold the NEOX pair (i, i + head_dim / 2) of their head (rows permuted at load); the reference build
                                // evaluates out[i] = fma(x0, cos, -(x1 * sin)), out[i + n/2] = fma(x0, sin, x1 * cos) (ggml.c:12543-12561, oracle/mirror.c:mir_rope_neox)
        const float o = (e_r & 1) ? fmaf(other, e_cs.y, res * e_cs.x) : fmaf(res, e_cs.x, -(other * e_cs.y));
        const int p = e_r % a.head_dim;
        const int r_orig = e_r - p + (p >> 1) + ((p & 1) ? (a.head_dim >> 1) : 0);
        if (e_epi == EPI_ROPE_Q) a.q_f16[r_orig] = f32_to_f16_bits(o);
        else a.kcache[kcache_off(pos, r_orig, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
    } else {   // EPI_ROPE_Q / EPI_ROPE_K: the pair (2u, 2u + 1) is one rotation (reference ggml.c:12536-12537, fma forms of the build)
        const float o = (e_r & 1) ? fmaf(res, e_cs.x, other * e_cs.y) : fmaf(res, e_cs.x, -(other * e_cs.y));
        if (e_epi == EPI_ROPE_Q) a.q_f16[e_r] = f32_to_f16_bits(o);
        else a.kcache[kcache_off(pos, e_r, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
    }
}

// TA / TB: weight types of the two job groups (TB == 0: one group).  Dynamic LDS: sizeof(SmemV9<MAXK>).
// NW waves per workgroup, NS ring slots per wave: the host launches (16, 3) for single-type K-quant launches and (16, 4) otherwise
// (engine.cc:launch_matvec_kq has the measurement).  Measured in round 4 (profiles/r04_*): NS = 6 / 8 at
// sixteen waves spill; (8, 11 | 8 | 7) — eight waves with 256 registers and a whole unit in flight, for launches in which a wave owns
// one unit of many records (ffn_down) — streams SLOWER (10.8 against 8.8 us per launch): a CU with eight streaming waves is served at
// about 8.6 B/cycle whatever they have in flight, with sixteen at 10.7, with no block math at 12.
// x0 / nw0 / K0 / pro0 repeat a.x, a.norm_w, a.K, a.pro as LEADING scalar arguments: the build preloads a kernel's first argument dwords
// into scalar registers at wave launch (-mllvm -amdgpu-kernarg-preload-count, Makefile), so the activation requests — the first thing every
// wave does — wait for no kernel-argument fetch (a by-value struct is not preloaded: its fields arrive by scalar loads, one or two cache-miss
// round trips after wave start).
template <int MAXK, int TA, int TB, bool LN, bool EMB = false, int NW = 16, int NS = 4>
__global__ void __launch_bounds__(64 * NW) matvec_v9_kernel(const float* x0, const float* nw0, int K0, int pro0, const MatvecArgs a) {
    // (preloading the arenas and unit counts too — what the first weight requests need, 13 dwords — measured 0.3 % slower than these 6)
    CT_DYN_SMEM(smem_raw);
    SmemV9<MAXK>& SM = *reinterpret_cast<SmemV9<MAXK>*>(smem_raw);
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    constexpr bool B32 = is_b32<TA>();
    static_assert(!B32 || TB == 0, "32-block types: single-type launches");
    // Every wave's activation requests are in the CU's memory pipeline before ANY wave requests weights: the pipeline serves a
    // CU's requests in order, and an activation load queued behind other waves' weight records waits for them to stream in from
    // HBM (measured: with the whole ring requested before the prologue, its end moved from cycle 7200 to 11600).  The barrier also
    // publishes the cleared arrival counter of the prologue.
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv;
    const int grid = (int)gridDim.x, bx = (int)blockIdx.x;
    if constexpr (B32) {
        Pro9b<MAXK> P;
        pro9b_load<MAXK>(P, x0, nw0, K0, pro0);
        kernarg_touch<24 + sizeof(MatvecArgs)>();
        if (a.dbg & 64) {   // measurement only (CT_AMD_DBG=64), as below
            if (a.bump && bx == 0 && threadIdx.x == 0) { a.bump[0] += 1; a.bump[1] += 1; a.bump[4 + a.n_ctx] += 1; }
            return;
        }
        __syncthreads();
        const unsigned long long t0 = trace ? clock64_dev() : 0ull;
        auto pro = [&](bool, unsigned long long (&)[4]) __attribute__((always_inline)) {
            pro9b_finish<MAXK, TA == GT_Q4_0, EMB>(SM.L, P, a.norm_w, a.norm_b, a.K, a.pro, a.eps, a.emb_out);
        };
        static_assert(NW == 16, "32-block types: the 16-wave form");
        { Items9 it{bx + grid * wv, grid * 16, a.n_pairs}; v9_run<TA, MAXK, false, NS, EMB>(a, SM, a.baseA, 0, it, lane, wv, pro); }
        if (trace) { tr[0] = t0; tr[6] = clock64_dev(); }
        if (a.bump && bx == 0 && threadIdx.x == 0) { a.bump[0] += 1; a.bump[1] += 1; a.bump[4 + a.n_ctx] += 1; }
        return;
    } else {
    Pro9<MAXK, TB == 0, NW> P;
    pro9_load<MAXK, TB == 0, NW>(P, x0, nw0, K0, pro0, wv, lane);
    kernarg_touch<24 + sizeof(MatvecArgs)>();   // (gpu.h) behind the activation requests: one round trip for every argument line
    if (a.dbg & 64) {   // measurement only (CT_AMD_DBG=64): the launch and its boundary without the kernel's work (the cursor still advances)
        if (a.bump && blockIdx.x == 0 && threadIdx.x == 0) { a.bump[0] += 1; a.bump[1] += 1; a.bump[4 + a.n_ctx] += 1; }
        return;
    }
    if (threadIdx.x == 0) SM.L.cnt = 0u;
    __syncthreads();   // (round 5: moved behind the waves' first weight requests — so that the first waves do not wait for the last wave's launch, ~2000
                       // cycles — the token rate DROPPED 3.4 %: 712 against 737 tok/s, alternating on one box.  The order this barrier enforces is worth more.)
    const unsigned long long t0 = trace ? clock64_dev() : 0ull;
    auto pro = [&](bool trc, unsigned long long (&ts)[4]) __attribute__((always_inline)) {
        pro9_finish<MAXK, LN, EMB, TB == 0, TA == GT_Q6_K || TB == GT_Q6_K, NW>(SM.L, P, a.norm_w, a.norm_b, a.K, a.pro, a.eps, a.emb_out, wv, lane, trc, ts);
    };
    if constexpr (TB != 0) {
        static_assert(NW == 16, "two-type launches: the 16-wave form");
        const int nwA = a.nwA;
        if (wv < nwA) { Items9 it{bx + grid * wv, grid * nwA, a.n_groupA}; v9_run<TA, MAXK, true, NS, false>(a, SM, a.baseA, 0, it, lane, wv, pro); }
        else { Items9 it{a.n_groupA + bx + grid * (wv - nwA), grid * (16 - nwA), a.n_pairs}; v9_run<TB, MAXK, true, NS, false>(a, SM, a.baseB, a.n_groupA, it, lane, wv, pro); }
    } else {
        { Items9 it{bx + grid * wv, grid * NW, a.n_pairs}; v9_run<TA, MAXK, false, NS, EMB>(a, SM, a.baseA, 0, it, lane, wv, pro); }
    }
    if (trace) { tr[0] = t0; tr[6] = clock64_dev(); }
    if (a.bump && bx == 0 && threadIdx.x == 0) { a.bump[0] += 1; a.bump[1] += 1; a.bump[4 + a.n_ctx] += 1; }   // no wave of this launch reads the cursor (host: kernels.h); [4 + n_ctx]: the token epoch (kernels_qa9.h)
    }
}

