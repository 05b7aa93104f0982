// Mat-vec, generation 8 (K-quants, K <= 12288): the integer dots on the matrix cores, weights read in file order.
//
// Why: generation 7 (kernels_v7.h) streams its records at whatever the memory system gives (tools/experiments/stream_probe:
// 7.0 TB/s on exactly its access pattern) and still ran at 3.3 TB/s.  SQ counters (profiles/r02_v7_sq_counters_7b_q4km.txt)
// say why: a VALU instruction occupies a SIMD for 4 cycles per wave, the block math is ~100 of them per 1152-byte record,
// so 16 waves per CU can unpack and dot4 at most ~11 B per cycle and CU — the HBM rate itself, with nothing left for the
// prologue, the chain and the tails.  The decode mat-vec is VALU-bound, not memory-bound.
//
// What the matrix cores can take over although only ONE activation vector exists: the reference's lane sums
//     sumi[l] = sum_{s<8} sc_s * d_s[l],     d_s[l] = sum_{e<4} q_s[4l+e] * a_s[4l+e]        (k_quants.c:2651-2720)
// need sixteen 4-element dots per 64 weights (two sub-blocks x eight l).  One v_mfma_i32_16x16x64_i8 produces exactly those
// for 16 weight rows at once when its 16 "token" columns are sixteen MASKED COPIES of the activations:
//     pattern p = (l, c2):  A[p][k] = a[k] if k lies in sub-block c2 of the pair, elements 4l..4l+3;  0 elsewhere
//     D[p][row] = d_{sub-block c2}[l] of that row.
// 15/16 of the multipliers work on zeros — they are idle otherwise — and what it removes from the VALU is every dot4, the
// transpose-reduce across lanes and, because the results land in the lane that holds the row's header (transposed issue,
// as in kernels_pfm.h), all cross-lane traffic of the block math.  Per 16 rows x 256 weights and lane: 24 VALU to split
// nibbles, 4 MFMA, ~30 VALU for scales/mins/conversions: ~33 per 2048 weights against ~100.
//
// Shape of the work:
//   * weights in LAYOUT_ROWS: row-major, a row = its blocks in file order (Q4_K / Q5_K: the file bytes unchanged — lane
//     (row n, q) of the MFMA's B operand reads 16 consecutive nibble bytes; Q6_K: 208-byte slots ql|qh|scales + the row's d
//     values behind them, because 210-byte blocks cannot be read with 16-byte loads).
//   * a tile = 16 rows (8 row pairs; gate/up: 8 gate rows + 8 up rows), dealt to workgroups in contiguous, byte-balanced
//     ranges per job.  The 16 waves of a workgroup split the K-blocks of a tile (wave w: blocks w, w + 16, ...), so
//     M = 4096 still gives every wave work; a wave's (tile, block) items form one flat sequence with a 2-deep register ring.
//   * the reference's f32 fma chain over the blocks — the only order-dependent part — is replayed per tile from LDS chain
//     records (s[2q], s[2q+1], d, prod, dm per lane and block) by a rotating duty wave, lagging one tile behind behind
//     monotonic LDS counters (generation 6's protocol), then reduced with the AVX tree and parked in LDS; all fused
//     epilogues (and every global store) run after the loop, their operands requested before it (generation 7's lesson:
//     a store or a conditional load inside the streaming loop makes hipcc wait vmcnt(0) at every step).
//   * activations: quantized once per workgroup (the Q8_K prologue, unchanged arithmetic) into an LDS image where every
//     dword sits in a 16-byte cell followed by three zero dwords: an A operand is ONE ds_read_b128 at a lane-constant
//     offset (the dword lands in vector slot l & 3, the neighbours are padding zeros), inactive lanes read a zero page.
//   * Q6_K's -32 offset costs no VALU: a second MFMA with a constant B operand (-32 in every byte) accumulates
//     -32 * sum(a) onto every row's results.
#pragma once
#include "kernels_v7.h"

constexpr int kV8MaxTiles = 16;    // tiles one workgroup may own in a launch (host grows the grid otherwise)
constexpr int kV8Zero = 1024;      // bytes of the zero page in front of the activation cells

enum { LAYOUT_ROWS_Q6_SLOT = 208 };

template <int TYPE> struct TileImg;
template <> struct TileImg<GT_Q4_K> { u32x4 hdr, w0, w1; };
template <> struct TileImg<GT_Q5_K> { u32x4 hdr, w0, w1, qh; };
template <> struct TileImg<GT_Q6_K> { u32x4 ql0, ql1, qh0, qh1, sc; uint32_t d; };

// A (tile, block) item of lane (n, q): `base` = row 0 of the tile (wave-uniform: the loads take it as their scalar base), `vrow` =
// this lane's row offset inside the tile (bytes), b = block.
DEV const uint8_t* lane_ptr(const uint8_t* base, uint32_t off) { return base + (size_t)off; }
template <int TYPE> DEV TileImg<TYPE> tile_load(const uint8_t* base, uint32_t vrow, int b, int q, int nb);
template <> DEV TileImg<GT_Q4_K> tile_load<GT_Q4_K>(const uint8_t* base, uint32_t vrow, int b, int q, int) {
    const uint8_t* p = base + (size_t)b * 144;
    TileImg<GT_Q4_K> R;
    R.hdr = ld_stream16(lane_ptr(p, vrow));
    R.w0 = ld_stream16(lane_ptr(p, vrow + 16 + 16 * q));
    R.w1 = ld_stream16(lane_ptr(p, vrow + 80 + 16 * q));
    return R;
}
template <> DEV TileImg<GT_Q5_K> tile_load<GT_Q5_K>(const uint8_t* base, uint32_t vrow, int b, int q, int) {
    const uint8_t* p = base + (size_t)b * 176;
    TileImg<GT_Q5_K> R;
    R.hdr = ld_stream16(lane_ptr(p, vrow));
    R.qh = ld_stream16(lane_ptr(p, vrow + 16 + 16 * (q & 1)));
    R.w0 = ld_stream16(lane_ptr(p, vrow + 48 + 16 * q));
    R.w1 = ld_stream16(lane_ptr(p, vrow + 112 + 16 * q));
    return R;
}
template <> DEV TileImg<GT_Q6_K> tile_load<GT_Q6_K>(const uint8_t* base, uint32_t vrow, int b, int q, int nb) {
    const uint8_t* p = base + (size_t)b * LAYOUT_ROWS_Q6_SLOT;
    TileImg<GT_Q6_K> R;
    R.ql0 = ld_stream16(lane_ptr(p, vrow + 16 * q));
    R.ql1 = ld_stream16(lane_ptr(p, vrow + 64 + 16 * q));
    R.qh0 = ld_stream16(lane_ptr(p, vrow + 128 + 16 * (q & 1)));
    R.qh1 = ld_stream16(lane_ptr(p, vrow + 160 + 16 * (q & 1)));
    R.sc = ld_stream16(lane_ptr(p, vrow + 192));
    R.d = *(const uint16_t*)lane_ptr(base + (size_t)nb * LAYOUT_ROWS_Q6_SLOT + 2 * b, vrow);   // the row's d values sit behind its slots
    return R;
}
template <int TYPE> DEV constexpr int rows_block_bytes() { return TYPE == GT_Q4_K ? 144 : (TYPE == GT_Q5_K ? 176 : LAYOUT_ROWS_Q6_SLOT); }

// Byte offsets (from the start of dynamic LDS) of the pieces of a generation-8 workgroup's LDS; nb = K / 256.
struct SmemV8 {
    int pa, yd, sb, red, ctr, tt, res, c4, cm, total;
};
CT_HD static inline SmemV8 smem_v8(int nb, int nbuf) {
    SmemV8 s;
    s.pa = kV8Zero;                              // cells: 16 bytes per activation dword
    s.yd = s.pa + nb * 1024 + 16;
    s.sb = s.yd + nb * 4;
    s.red = (s.sb + nb * 32 + 15) & ~15;
    s.ctr = s.red + 128;                         // arrive[4] | freed[4]
    s.tt = s.ctr + 32;                           // tile table: kV8MaxTiles x TileRec8
    s.res = s.tt + kV8MaxTiles * 32;
    s.c4 = s.res + kV8MaxTiles * 16 * 4;         // chain records: float4 per (slot, block, lane)
    s.cm = s.c4 + nbuf * nb * 1024;              //                float  per (slot, block, lane)
    s.total = s.cm + nbuf * nb * 256;
    return s;
}

// Prologue part 2 for generation 8: pro7_finish's arithmetic, the image in the padded-cell form.
template <int MAXK, bool LN, bool EMB>
DEV void pro8_finish(unsigned char* sm, const SmemV8& S, ProRegs7<MAXK>& P, const float* __restrict__ nw, const float* __restrict__ nbias,
                     int K, int pro, float eps, float* __restrict__ emb_out) {
    constexpr int ROUNDS = ProRegs7<MAXK>::ROUNDS, NW = 16;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, sub = tid & 15, grp = tid >> 4;
    const int nblk = K >> 8;
    double* red = (double*)(sm + S.red);
    float* ydp = (float*)(sm + S.yd);
    int* sbp = (int*)(sm + S.sb);
    if (tid < 256) ((int*)sm)[tid] = 0;   // the zero page
    const bool wave_live = uniform_int(wv * 4) < nblk;   // dead waves skip the arithmetic (wave-uniform branches)
    float scale = 1.0f;
    if (pro == PRO_RMSNORM) {
        double s = 0.0;
        if (wave_live) {
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                if (grp + rd * 64 < nblk) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s += (double)(P.v[rd][k].x * P.v[rd][k].x);
                        s += (double)(P.v[rd][k].y * P.v[rd][k].y);
                        s += (double)(P.v[rd][k].z * P.v[rd][k].z);
                        s += (double)(P.v[rd][k].w * P.v[rd][k].w);
                    }
                }
            }
            s = wave_sum_fast(s);
        }
        if (lane == 0) red[wv] = wave_live ? s : 0.0;
        __syncthreads();
        if (wave_live) {
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += red[w];
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + eps);
        }
    }
    if constexpr (LN) {
        if (pro == PRO_LAYERNORM) {
            double s1 = 0.0;
            if (wave_live) {
#pragma unroll
                for (int rd = 0; rd < ROUNDS; ++rd) {
                    if (grp + rd * 64 < nblk) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            s1 += (double)P.v[rd][k].x; s1 += (double)P.v[rd][k].y; s1 += (double)P.v[rd][k].z; s1 += (double)P.v[rd][k].w;
                        }
                    }
                }
                s1 = wave_sum_fast(s1);
            }
            if (lane == 0) red[wv] = wave_live ? s1 : 0.0;
            __syncthreads();
            double tot = 0.0;
            for (int w = 0; w < NW; ++w) tot += red[w];
            const float mean = (float)(tot / (double)K);
            __syncthreads();   // red is reused for the second moment
            double s2 = 0.0;
            if (wave_live) {
#pragma unroll
                for (int rd = 0; rd < ROUNDS; ++rd) {
                    if (grp + rd * 64 < nblk) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float4& q = P.v[rd][k];
                            q.x -= mean; q.y -= mean; q.z -= mean; q.w -= mean;
                            s2 += (double)(q.x * q.x); s2 += (double)(q.y * q.y); s2 += (double)(q.z * q.z); s2 += (double)(q.w * q.w);
                        }
                    }
                }
                s2 = wave_sum_fast(s2);
            }
            if (lane == 0) red[wv] = wave_live ? s2 : 0.0;
            __syncthreads();
            double tot2 = 0.0;
            for (int w = 0; w < NW; ++w) tot2 += red[w];
            const float variance = (float)(tot2 / (double)K);
            scale = 1.0f / sqrtf(variance + eps);
        }
    }
    if (wave_live) {
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int b = grp + rd * 64;
            const bool live = b < nblk;            // uniform within a 16-lane row, may differ between rows of a wave
            float t[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 q = live ? P.v[rd][k] : float4{0.f, 0.f, 0.f, 0.f};
                if (live && pro != PRO_PLAIN) {
                    float4 w4;
                    if constexpr (ProRegs7<MAXK>::EARLY_W) w4 = P.w[rd][k];
                    else w4 = *(const float4*)(nw + b * 256 + sub * 16 + k * 4);
                    q.x = (q.x * scale) * w4.x;
                    q.y = (q.y * scale) * w4.y;
                    q.z = (q.z * scale) * w4.z;
                    q.w = (q.w * scale) * w4.w;
                    if constexpr (LN) {
                        if (pro == PRO_LAYERNORM) {
                            const float4 b4 = *(const float4*)(nbias + b * 256 + sub * 16 + k * 4);
                            q.x += b4.x; q.y += b4.y; q.z += b4.z; q.w += b4.w;
                        }
                    }
                    if constexpr (EMB) {   // a store here is a pending write at the entry of the streaming loop: lm_head instantiation only
                        if (emb_out && blockIdx.x == 0) *(float4*)(emb_out + b * 256 + sub * 16 + k * 4) = q;
                    }
                }
                t[4 * k] = q.x; t[4 * k + 1] = q.y; t[4 * k + 2] = q.z; t[4 * k + 3] = q.w;
            }
            float am = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(t[e]));
            float amax = am;
            amax = fmaxf(amax, lane_xor1(amax));
            amax = fmaxf(amax, lane_xor2(amax));
            amax = fmaxf(amax, lane_xor4(amax));
            amax = fmaxf(amax, lane_xor8(amax));
            // first element (lowest index) attaining amax keeps its sign
            const unsigned long long hit = __ballot(am == amax);
            const unsigned row_bits = (unsigned)((hit >> (lane & 48)) & 0xFFFFu);
            const int first = (lane & 48) + (__ffsll((unsigned long long)row_bits) - 1);
            float mine = 0.0f;
#pragma unroll
            for (int e = 15; e >= 0; --e) mine = (fabsf(t[e]) == amax) ? t[e] : mine;
            const float maxv = __shfl(mine, first);
            int packed[4] = {0, 0, 0, 0}, s16 = 0;
            float d = 0.0f;
            if (amax != 0.0f) {
                const float iscale = -128.f / maxv;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int q = ((int)f32_to_bits(fmaf(iscale, t[e], 12582912.f)) & 0x007fffff) - 0x00400000;
                    q = q > 127 ? 127 : q;
                    packed[e >> 2] |= (q & 0xff) << (8 * (e & 3));
                    s16 += q;
                }
                d = 1.0f / iscale;
            }
            const int s32 = s16 + lane_xor1(s16);
            if (live) {
                unsigned char* cell = sm + S.pa + (size_t)(b * 64 + sub * 4) * 16;   // dword k of this lane -> cell b*64 + 4*sub + k
#pragma unroll
                for (int k = 0; k < 4; ++k) *(u32x4*)(cell + 16 * k) = u32x4{(uint32_t)packed[k], 0u, 0u, 0u};
                if ((sub & 1) == 0) sbp[b * 8 + (sub >> 1)] = s32;
                if (sub == 0) ydp[b] = d;
            }
        }
    }
    __syncthreads();
}

// The 8 sub-block scales / mins of a Q4_K / Q5_K header as bytes of two dwords each (reference get_scale_min_k4,
// k_quants.c:306-314: j < 4: q[j] & 63, q[j+4] & 63; else (q[j+4] & 0xF) | ((q[j-4] >> 6) << 4), (q[j+4] >> 4) | ((q[j] >> 6) << 4)).
DEV void k4_scales(const u32x4 hdr, uint32_t& sc03, uint32_t& sc47, uint32_t& m03, uint32_t& m47) {
    const uint32_t A = hdr[1], B = hdr[2], C = hdr[3];
    sc03 = A & 0x3F3F3F3Fu;
    sc47 = (C & 0x0F0F0F0Fu) | ((A >> 2) & 0x30303030u);
    m03 = B & 0x3F3F3F3Fu;
    m47 = ((C >> 4) & 0x0F0F0F0Fu) | ((B >> 2) & 0x30303030u);
}
DEV int byte_of(uint32_t w, int k) { return (int)((w >> (8 * k)) & 0xFFu); }

// Per-lane constants of a generation-8 wave.
struct Geom8 {
    int n, q;          // weight row inside the tile, k-quarter
    int a_off;         // byte offset (from the start of LDS) of this lane's A-operand window in block 0, variant offset 0
                       // (inactive lanes: 0 = the zero page)
    int a_blk;         // 1024 for active lanes, 0 for inactive ones: what a block adds to a_off
};
// TYPE-dependent: distance (in sub-blocks of 32) between the two sub-blocks of an MFMA's k range (c2 = 0 / 1).
template <int TYPE> DEV Geom8 geom8(int lane) {
    Geom8 g;
    g.n = lane & 15;
    g.q = lane >> 4;
    const int p = lane & 15, l = p >> 1, c2 = p & 1;
    const bool active = g.q == 2 * c2 + (l >> 2);
    constexpr int CS = TYPE == GT_Q6_K ? 1 : 2;
    g.a_off = active ? kV8Zero + (CS * c2 * 8 + l) * 16 - 4 * (l & 3) : 0;
    g.a_blk = active ? 1024 : 0;
    return g;
}
DEV u32x4 a_operand(const unsigned char* sm, const Geom8& g, int b, int s0) {   // s0: first sub-block of the variant (k range of c2 = 0)
    return *(const u32x4*)(sm + g.a_off + g.a_blk * b + 128 * s0);
}

// One (tile, block) item of this wave -> the chain record of its lane: s0, s1 = (float)sumi[2q], (float)sumi[2q+1]; d = y.d * d_row;
// p = (float)prod (Q4_K: prod[t = q]; Q5_K: the block total); dm = -y.d * dmin_row.
template <int TYPE>
DEV void tile_math(const TileImg<TYPE>& R, const unsigned char* sm, const SmemV8& S, const Geom8& g, int b, float& s0, float& s1, float& d,
                   float& p, float& dm) {
    const float yd = ((const float*)(sm + S.yd))[b];
    const i32x4 zero = {0, 0, 0, 0};
    const int q = g.q;
    if constexpr (TYPE == GT_Q4_K || TYPE == GT_Q5_K) {
        uint32_t lo0[4], hi0[4], lo1[4], hi1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo0[k] = R.w0[k] & 0x0F0F0F0Fu; hi0[k] = (R.w0[k] >> 4) & 0x0F0F0F0Fu;
            lo1[k] = R.w1[k] & 0x0F0F0F0Fu; hi1[k] = (R.w1[k] >> 4) & 0x0F0F0F0Fu;
        }
        if constexpr (TYPE == GT_Q5_K) {   // fifth bit: byte e of qh, bit s of sub-block s; this lane's sub-blocks: 2*(2*m2 + (q >> 1)) + nib
            const int sh = 2 * (q >> 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo0[k] |= ((R.qh[k] >> sh) & 0x01010101u) << 4;
                hi0[k] |= ((R.qh[k] >> (sh + 1)) & 0x01010101u) << 4;
                lo1[k] |= ((R.qh[k] >> (sh + 4)) & 0x01010101u) << 4;
                hi1[k] |= ((R.qh[k] >> (sh + 5)) & 0x01010101u) << 4;
            }
        }
        // D[v][r]: pattern 4q + r = (l = 2q + (r >> 1), c2 = r & 1); sub-block of (v, c2): v0 -> 0/2, v1 -> 1/3, v2 -> 4/6, v3 -> 5/7
        const i32x4 D0 = mfma_i8_16x16x64(a_operand(sm, g, b, 0), u32x4{lo0[0], lo0[1], lo0[2], lo0[3]}, zero);
        const i32x4 D1 = mfma_i8_16x16x64(a_operand(sm, g, b, 1), u32x4{hi0[0], hi0[1], hi0[2], hi0[3]}, zero);
        const i32x4 D2 = mfma_i8_16x16x64(a_operand(sm, g, b, 4), u32x4{lo1[0], lo1[1], lo1[2], lo1[3]}, zero);
        const i32x4 D3 = mfma_i8_16x16x64(a_operand(sm, g, b, 5), u32x4{hi1[0], hi1[1], hi1[2], hi1[3]}, zero);
        uint32_t sc03, sc47, m03, m47;
        k4_scales(R.hdr, sc03, sc47, m03, m47);
        const int c0 = byte_of(sc03, 0), c1 = byte_of(sc03, 1), c2s = byte_of(sc03, 2), c3 = byte_of(sc03, 3);
        const int c4 = byte_of(sc47, 0), c5 = byte_of(sc47, 1), c6 = byte_of(sc47, 2), c7 = byte_of(sc47, 3);
        const int sum0 = mul24(c0, D0[0]) + mul24(c2s, D0[1]) + mul24(c1, D1[0]) + mul24(c3, D1[1]) + mul24(c4, D2[0]) + mul24(c6, D2[1]) +
                         mul24(c5, D3[0]) + mul24(c7, D3[1]);
        const int sum1 = mul24(c0, D0[2]) + mul24(c2s, D0[3]) + mul24(c1, D1[2]) + mul24(c3, D1[3]) + mul24(c4, D2[2]) + mul24(c6, D2[3]) +
                         mul24(c5, D3[2]) + mul24(c7, D3[3]);
        s0 = (float)sum0;
        s1 = (float)sum1;
        const int* sbp = (const int*)(sm + S.sb) + b * 8;
        if constexpr (TYPE == GT_Q4_K) {   // prod[t = q] = m[2q] * q8s[2q] + m[2q+1] * q8s[2q+1]
            const uint32_t mm = q < 2 ? m03 : m47;
            const int sh = 16 * (q & 1);
            const int ma = (int)((mm >> sh) & 0xFFu), mb = (int)((mm >> (sh + 8)) & 0xFFu);
            p = (float)(mul24(ma, sbp[2 * q]) + mul24(mb, sbp[2 * q + 1]));
        } else {                           // one scalar chain on the block total (k_quants.c:3183-3262)
            int tot = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) tot += mul24(byte_of(m03, j), sbp[j]) + mul24(byte_of(m47, j), sbp[4 + j]);
            p = (float)tot;
        }
        d = yd * f16_bits_to_f32((uint16_t)(R.hdr[0] & 0xFFFF));
        dm = -yd * f16_bits_to_f32((uint16_t)(R.hdr[0] >> 16));
    } else {   // Q6_K: sub-vectors of 32 in two halves of 128; variant a = low nibbles (sub-vectors 4j + c2), b = high nibbles (4j + 2 + c2)
        const int sh = 2 * (q >> 1);
        uint32_t a0[4], b0[4], a1[4], b1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a0[k] = (R.ql0[k] & 0x0F0F0F0Fu) | (((R.qh0[k] >> sh) & 0x03030303u) << 4);
            b0[k] = ((R.ql0[k] >> 4) & 0x0F0F0F0Fu) | (((R.qh0[k] >> (sh + 4)) & 0x03030303u) << 4);
            a1[k] = (R.ql1[k] & 0x0F0F0F0Fu) | (((R.qh1[k] >> sh) & 0x03030303u) << 4);
            b1[k] = ((R.ql1[k] >> 4) & 0x0F0F0F0Fu) | (((R.qh1[k] >> (sh + 4)) & 0x03030303u) << 4);
        }
        const u32x4 m32 = {0xE0E0E0E0u, 0xE0E0E0E0u, 0xE0E0E0E0u, 0xE0E0E0E0u};   // -32 in every byte: (q6 - 32) . a = q6 . a + (-32) . a
        u32x4 A;
        A = a_operand(sm, g, b, 0); const i32x4 D0 = mfma_i8_16x16x64(A, m32, mfma_i8_16x16x64(A, u32x4{a0[0], a0[1], a0[2], a0[3]}, zero));
        A = a_operand(sm, g, b, 2); const i32x4 D1 = mfma_i8_16x16x64(A, m32, mfma_i8_16x16x64(A, u32x4{b0[0], b0[1], b0[2], b0[3]}, zero));
        A = a_operand(sm, g, b, 4); const i32x4 D2 = mfma_i8_16x16x64(A, m32, mfma_i8_16x16x64(A, u32x4{a1[0], a1[1], a1[2], a1[3]}, zero));
        A = a_operand(sm, g, b, 6); const i32x4 D3 = mfma_i8_16x16x64(A, m32, mfma_i8_16x16x64(A, u32x4{b1[0], b1[1], b1[2], b1[3]}, zero));
        // scale of (sub-vector sv, l): sc[2 * sv + (l >> 2)], l >> 2 == q >> 1 for this lane's l = 2q, 2q + 1; sv of (v, c2) = S0_v + c2
        const int bs = 8 * (q >> 1);
        auto scl = [&](int sv) __attribute__((always_inline)) { return (int)(int8_t)((R.sc[sv >> 1] >> (16 * (sv & 1) + bs)) & 0xFFu); };
        const int e0 = scl(0), e1 = scl(1), e2 = scl(2), e3 = scl(3), e4 = scl(4), e5 = scl(5), e6 = scl(6), e7 = scl(7);
        const int sum0 = mul24(e0, D0[0]) + mul24(e1, D0[1]) + mul24(e2, D1[0]) + mul24(e3, D1[1]) + mul24(e4, D2[0]) + mul24(e5, D2[1]) +
                         mul24(e6, D3[0]) + mul24(e7, D3[1]);
        const int sum1 = mul24(e0, D0[2]) + mul24(e1, D0[3]) + mul24(e2, D1[2]) + mul24(e3, D1[3]) + mul24(e4, D2[2]) + mul24(e5, D2[3]) +
                         mul24(e6, D3[2]) + mul24(e7, D3[3]);
        s0 = (float)sum0;
        s1 = (float)sum1;
        d = yd * f16_bits_to_f32((uint16_t)(R.d & 0xFFFF));
        p = 0.0f;
        dm = 0.0f;
    }
}

// A tile of the calling workgroup, as the kernel's first 16 threads leave it in LDS for everybody (computed once per launch:
// the streaming loop then never looks at the job table again).
struct TileRec8 {
    long long off;         // row 0 of the tile in its matrix (gate/up: in the gate matrix), as a byte offset from job 0's rows: a
                           // pointer that went through LDS comes back as a FLAT pointer — its loads then also count on lgkmcnt and
                           // every LDS wait of the loop waits for the weight stream; base + offset keeps them global loads
    int nvalid;            // rows of the tile that exist (plain: <= 16, gate/up: <= 8 of each matrix)
    int job;
    int row0;              // first output row
    int pad[3];
};

struct Jobs8 {      // the launch's jobs in scalars.  No arrays: hipcc turns a select chain over array elements into a dynamically
                    // indexed stack object, i.e. scratch loads (which count on vmcnt) in the middle of the streaming loop.
    int lo0, lo1, lo2;        // first row pair of this workgroup's range in job j
    int tb1, tb2, tb3;        // tiles of this workgroup before job 1, before job 2, in total
};
// Job j's row pairs are dealt to the workgroups in contiguous ranges: the first `pr` workgroups get `pq + 1` pairs, the others `pq`
// (host: pq = pairs / grid, pr = pairs % grid — no division on the device).
DEV int pair_lo(int pq, int pr, int bx) { return bx * pq + (bx < pr ? bx : pr); }
DEV int pair_cnt(int pq, int pr, int bx) { return pq + (bx < pr ? 1 : 0); }
DEV Jobs8 jobs8(const MatvecArgs& a) {
    Jobs8 J;
    const int bx = (int)blockIdx.x;
    J.lo0 = pair_lo(a.job[0].pq, a.job[0].pr, bx);
    J.lo1 = a.njobs > 1 ? pair_lo(a.job[1].pq, a.job[1].pr, bx) : 0;
    J.lo2 = a.njobs > 2 ? pair_lo(a.job[2].pq, a.job[2].pr, bx) : 0;
    const int c0 = (pair_cnt(a.job[0].pq, a.job[0].pr, bx) + 7) / 8;
    const int c1 = a.njobs > 1 ? (pair_cnt(a.job[1].pq, a.job[1].pr, bx) + 7) / 8 : 0;
    const int c2 = a.njobs > 2 ? (pair_cnt(a.job[2].pq, a.job[2].pr, bx) + 7) / 8 : 0;
    J.tb1 = c0; J.tb2 = c0 + c1; J.tb3 = c0 + c1 + c2;
    J.lo0 = uniform_int(J.lo0); J.lo1 = uniform_int(J.lo1); J.lo2 = uniform_int(J.lo2);
    J.tb1 = uniform_int(J.tb1); J.tb2 = uniform_int(J.tb2); J.tb3 = uniform_int(J.tb3);
    return J;
}
// Tile i of this workgroup (any thread; used once per tile at kernel start).
DEV TileRec8 tile_rec(const MatvecArgs& a, const Jobs8 J, int i) {
    const bool j2 = i >= J.tb2, j1 = !j2 && i >= J.tb1;
    const int k = i - (j2 ? J.tb2 : (j1 ? J.tb1 : 0));
    const int lo = j2 ? J.lo2 : (j1 ? J.lo1 : J.lo0);
    const uint8_t* base = j2 ? a.job[2].w.rows : (j1 ? a.job[1].w.rows : a.job[0].w.rows);
    const int M = j2 ? a.job[2].w.M : (j1 ? a.job[1].w.M : a.job[0].w.M);
    const int rb = j2 ? a.job[2].w.row_bytes : (j1 ? a.job[1].w.row_bytes : a.job[0].w.row_bytes);
    const int cnt = pair_cnt(j2 ? a.job[2].pq : (j1 ? a.job[1].pq : a.job[0].pq), j2 ? a.job[2].pr : (j1 ? a.job[1].pr : a.job[0].pr), (int)blockIdx.x);
    const int pair0 = lo + 8 * k;
    const int np = cnt - 8 * k < 8 ? cnt - 8 * k : 8;
    TileRec8 t;
    t.job = j2 ? 2 : (j1 ? 1 : 0);
    t.row0 = a.gateup ? pair0 : 2 * pair0;
    int nv = a.gateup ? np : 2 * np;
    nv = nv < M - t.row0 ? nv : M - t.row0;
    t.nvalid = nv > 0 ? nv : 1;
    t.off = (long long)(base - a.job[0].w.rows) + (long long)t.row0 * (long long)rb;
    t.pad[0] = t.pad[1] = t.pad[2] = 0;
    return t;
}

// The chain replay of one tile by its duty wave (reference order: one fma per block in block order, then hsum_float_8 and the
// min-term tree, kernels_exact.h), result of row n -> RES[16 * i + n].  chain0 = first chain record of the slot (block 0, lane 0).
template <int TYPE>
DEV void tile_replay(unsigned char* sm, const SmemV8& S, int chain0, int nb, int i, int lane) {
    constexpr bool mins = TYPE != GT_Q6_K;
    const float4* c4 = (const float4*)(sm + S.c4) + chain0 + lane;
    const float* cm = (const float*)(sm + S.cm) + chain0 + lane;
    float acc0 = 0.0f, acc1 = 0.0f, accm = 0.0f;
    int b = 0;
    for (; b + 8 <= nb; b += 8) {   // operands of 8 blocks are fetched before their dependent fmas
        float4 v[8];
        float m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = c4[(b + u) * 64];
            if constexpr (mins) m[u] = cm[(b + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = fmaf(v[u].z, v[u].x, acc0);
            acc1 = fmaf(v[u].z, v[u].y, acc1);
            if constexpr (mins) accm = fmaf(m[u], v[u].w, accm);
        }
    }
    for (; b < nb; ++b) {
        const float4 v = c4[b * 64];
        acc0 = fmaf(v.z, v.x, acc0);
        acc1 = fmaf(v.z, v.y, acc1);
        if constexpr (mins) accm = fmaf(cm[b * 64], v.w, accm);
    }
    // x_l = chain l of the row: l = 2q + j.  hsum_float_8 = ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7)): lanes q and q ^ 2 are 32 apart,
    // q and q ^ 1 are 16 apart.
    const float e02 = acc0 + lane_xor32(acc0), o02 = acc1 + lane_xor32(acc1);
    const float ee = e02 + lane_xor16(e02), oo = o02 + lane_xor16(o02);
    float res = ee + oo;
    if constexpr (TYPE == GT_Q4_K) {        // (m0 + m2) + (m1 + m3), accumulator t in lane q = t
        const float m02 = accm + lane_xor32(accm);
        res = res + (m02 + lane_xor16(m02));
    } else if constexpr (TYPE == GT_Q5_K) {
        res = res + accm;
    }
    if (lane < 16) ((float*)(sm + S.res))[16 * i + lane] = res;
}

struct Stamps8 { unsigned long long first_math = 0, first_arrive = 0, loop_end = 0, duty_end = 0; };

// Where the chain protocol stands (wave-uniform; carried from phase to phase): tile index, chain slot, how often the slot was used
// before, and which wave replays the tile.  Advanced incrementally: no division in the loop.
struct Chain8 {
    int i, slot, gen, duty;
};
DEV void chain_next(Chain8& c, int nbuf, int nwork) {
    ++c.i;
    if (++c.slot == nbuf) { c.slot = 0; ++c.gen; }
    if (++c.duty == nwork) c.duty = 0;
}

// All tiles [C.i, i1) of one weight type.  base0 / nvalid0: the phase's first tile (the tile table in LDS is not there yet when the
// first phase requests its first item).  `hook` runs once, after this wave's first item is requested: the kernel puts the prologue
// there (its latency then overlaps the first weights in flight; for a second phase it is empty).
template <int TYPE, class Hook>
DEV void v8_phase(const MatvecArgs& a, unsigned char* sm, const SmemV8 S, Chain8& C, int i1, const uint8_t* base0, int nvalid0, int rb, int lane,
                  int wv, Hook hook, Stamps8& ts, bool trace) {
    const int nb = a.K >> 8, nbuf = a.nbuf;
    const int nwork = nb < 16 ? nb : 16;            // waves that own blocks
    const int nbw = wv < nb ? (nb - wv + 15) / 16 : 0;   // blocks of this wave per tile: wv, wv + 16, ...
    const Geom8 g = geom8<TYPE>(lane);
    unsigned* arrive = (unsigned*)(sm + S.ctr);
    unsigned* freed = arrive + 4;
    unsigned char* c4b = sm + S.c4 + lane * 16;
    unsigned char* cmb = sm + S.cm + lane * 4;
    const TileRec8* TT = (const TileRec8*)(sm + S.tt);
    const int i0 = C.i, nt = i1 - i0;
    const int total = nt * nbw;
    // this lane's row offset inside a tile: row n (gate/up: row n & 7 of the gate matrix, lanes 8..15 of the up matrix)
    const bool gu = a.gateup != 0;
    const int full = gu ? 8 : 16;
    const int n_loc = gu ? (g.n & 7) : g.n;
    const uint32_t gu_off = (gu && g.n >= 8) ? (uint32_t)a.up_delta : 0u;
    const uint32_t v_full = (uint32_t)n_loc * (uint32_t)rb + gu_off;
    auto vrow_of = [&](int nvalid) __attribute__((always_inline)) {   // rows past the tile's end read its last row (results dropped)
        if (nvalid == full) return v_full;
        const int ne = n_loc < nvalid ? n_loc : nvalid - 1;
        return (uint32_t)ne * (uint32_t)rb + gu_off;
    };
    // ---- prefetch cursor: (pf_i, pf_j) is the item requested next; advanced lazily (the table may only be read after the prologue) ----
    int pf_i = i0, pf_j = 0, pf_left = total;
    const uint8_t* pf_base = base0;
    uint32_t pf_v = vrow_of(nvalid0);
    bool pf_adv = false;
    TileImg<TYPE> ring[2];
    auto issue = [&](TileImg<TYPE>& slot) __attribute__((always_inline)) {
        if (pf_adv) {
            if (++pf_j == nbw) {
                pf_j = 0;
                ++pf_i;
                const TileRec8 t = TT[pf_i];
                const long long off = ((long long)uniform_int((int)(t.off >> 32)) << 32) | (long long)(unsigned)uniform_int((int)(t.off & 0xffffffffll));
                pf_base = a.job[0].w.rows + off;
                pf_v = vrow_of(uniform_int(t.nvalid));
            }
        }
        slot = tile_load<TYPE>(pf_base, pf_v, wv + 16 * pf_j, g.q, nb);   // unconditional; past the end: the last item again
        pf_adv = pf_left > 1;
        pf_left -= pf_adv ? 1 : 0;
    };
    if (nbw > 0 && nt > 0) issue(ring[0]);   // ONE item per wave (37 KB per CU) before the prologue: more only blocks its barriers
    hook();
    if (nbw > 0 && nt > 0) issue(ring[1]);
    // Chain duty runs `lag` tiles behind the block math: with two or more chain slots the replay of tile i - 1 overlaps everybody's
    // math of tile i; with one slot nobody may start tile i before tile i - 1 is replayed, so the duty follows at once.
    const bool lag = nbuf > 1;
    Chain8 prev = C;
    bool have_prev = false;
    auto duty = [&](const Chain8& c) __attribute__((always_inline)) {
        if (wv == c.duty) {
            lds_wait_ge(&arrive[c.slot], (unsigned)nwork * (unsigned)(c.gen + 1));
            tile_replay<TYPE>(sm, S, c.slot * nb * 64, nb, c.i, lane);
            lds_signal(&freed[c.slot], lane, 1u);
        }
    };
    int jj = 0;
    for (int st = 0; st < total; st += 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool real = st + k < total;   // the surplus step of an odd item count re-processes the last item: results dropped
            if (jj == 0 && real) lds_wait_ge(&freed[C.slot], (unsigned)C.gen);   // the slot's previous tile was replayed
            const int b = wv + 16 * jj;
            float s0, s1, d, p, dm;
            tile_math<TYPE>(ring[k], sm, S, g, b, s0, s1, d, p, dm);
            reg_fence(s0, s1, d, p);
            if (trace && ts.first_math == 0) ts.first_math = clock64_dev();
            issue(ring[k]);
            if (real) {
                const int rec = (C.slot * nb + b) * 64;
                *(float4*)(c4b + (size_t)rec * 16) = float4{s0, s1, d, p};
                if constexpr (TYPE != GT_Q6_K) *(float*)(cmb + (size_t)rec * 4) = dm;
            }
            if (jj + 1 < nbw) { ++jj; continue; }
            if (real) {
                lds_signal(&arrive[C.slot], lane, 1u);
                if (trace && ts.first_arrive == 0) ts.first_arrive = clock64_dev();
                if (lag) { if (have_prev) duty(prev); }
                else duty(C);
                prev = C;
                have_prev = true;
                chain_next(C, nbuf, nwork);
                jj = 0;
            }
        }
    }
    if (trace) ts.loop_end = clock64_dev();
    if (nbw == 0 && nt > 0) {   // a wave without blocks still follows the tile sequence (its state feeds the next phase)
        for (int t = 0; t < nt; ++t) chain_next(C, nbuf, nwork);
    } else if (lag && have_prev) {
        duty(prev);   // the phase's last tile
    }
    if (trace) ts.duty_end = clock64_dev();
}

// TA / TB: weight types of the two job groups (TB == 0: one group); jobs of type TA come first.  Dynamic LDS: smem_v8(nb, nbuf).total.
template <int MAXK, int TA, int TB, bool LN, bool EMB = false>
__global__ void __launch_bounds__(1024) matvec_v8_kernel(const MatvecArgs a) {
    CT_DYN_SMEM(sm);
    ProRegs7<MAXK> P;
    pro7_load<MAXK>(P, a.x, a.norm_w, a.K, a.pro);
    const int lane = lane_id();
    const int wv = uniform_int(wave_id());
    const bool trace = (a.dbg & 32) && blockIdx.x == 0 && lane == 0;
    unsigned long long* tr = (unsigned long long*)a.dbg_sink + 16 * wv;
    const unsigned long long t0 = trace ? clock64_dev() : 0ull;
    const int nb = a.K >> 8;
    const SmemV8 S = smem_v8(nb, a.nbuf);
    const Jobs8 J = jobs8(a);
    const int T = J.tb3;
    const int tid = (int)threadIdx.x;
    if (tid < 8) ((unsigned*)(sm + S.ctr))[tid] = 0u;                              // published by the prologue's barriers
    if (tid < kV8MaxTiles && tid < T) ((TileRec8*)(sm + S.tt))[tid] = tile_rec(a, J, tid);   // likewise
    // the first tile in scalars: its first item is requested before the prologue (job 0, first pair of this workgroup's range)
    const int rbA = a.job[0].w.row_bytes;
    int nvalid0;
    {
        const int cnt0 = pair_cnt(a.job[0].pq, a.job[0].pr, (int)blockIdx.x), np = cnt0 < 8 ? cnt0 : 8;
        const int row0 = a.gateup ? J.lo0 : 2 * J.lo0;
        int nv = a.gateup ? np : 2 * np;
        nv = nv < a.job[0].w.M - row0 ? nv : a.job[0].w.M - row0;
        nvalid0 = nv > 0 ? nv : 1;
    }
    const uint8_t* base0 = a.job[0].w.rows + (size_t)(a.gateup ? J.lo0 : 2 * J.lo0) * (size_t)rbA;
    // ---- epilogue bookkeeping of thread t = (tile t >> 4, row t & 15); filled in by the hook ----
    const int e_i = tid >> 4, e_n = tid & 15;
    int e_r = 0, e_epi = 0, pos = 0;
    bool e_own = false;
    float e_res = 0.0f, e_res2 = 0.0f;
    float2 e_cs = float2{0.0f, 0.0f};
    unsigned long long t2 = 0ull;
    bool need_pos = false;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) need_pos = need_pos || (jj < a.njobs && (a.job[jj].epi == EPI_ROPE_Q || a.job[jj].epi == EPI_ROPE_K || a.job[jj].epi == EPI_V));
    // prologue + the epilogue operands (requested before the streaming loop, whose body must not contain a load besides the ring's);
    // runs inside the first phase, after its first weight request
    auto hook = [&]() __attribute__((always_inline)) {
        pro8_finish<MAXK, LN, EMB>(sm, S, P, a.norm_w, a.norm_b, a.K, a.pro, a.eps, a.emb_out);
        t2 = trace ? clock64_dev() : 0ull;
        if (e_i < T && e_i < kV8MaxTiles) {
            const TileRec8 t = ((const TileRec8*)(sm + S.tt))[e_i];
            e_epi = t.job == 2 ? a.job[2].epi : (t.job == 1 ? a.job[1].epi : a.job[0].epi);
            if (a.gateup) { e_r = t.row0 + (e_n & 7); e_own = e_n < 8 && e_n < t.nvalid; }
            else { e_r = t.row0 + e_n; e_own = e_n < t.nvalid; }
        }
        const bool need_res = e_own && (e_epi == EPI_ADD || e_epi == EPI_ADD2), need_res2 = e_own && e_epi == EPI_ADD2;
        const bool need_rope = e_own && (e_epi == EPI_ROPE_Q || e_epi == EPI_ROPE_K);
        pos = (need_pos && a.pos) ? sload_i32(a.pos) : 0;
        e_res = (need_res ? a.res : a.x)[need_res ? e_r : 0];
        e_res2 = (need_res2 ? a.res2 : a.x)[need_res2 ? e_r : 0];
        e_cs = *(const float2*)((need_rope ? a.rope_cs : a.x) +
                                (need_rope ? ((size_t)pos * (a.head_dim >> 1) + ((e_r % a.head_dim) >> 1)) * 2 : 0));
    };
    // ---- the tiles ----
    Stamps8 ts;
    Chain8 C;
    C.i = 0; C.slot = 0; C.gen = 0; C.duty = 0;
    if constexpr (TB != 0) {
        const int na = a.n_groupA;   // jobs of type TA
        const int ia = na == 0 ? 0 : (na == 1 ? J.tb1 : (na == 2 ? J.tb2 : J.tb3));
        v8_phase<TA>(a, sm, S, C, ia, base0, nvalid0, rbA, lane, wv, hook, ts, trace);
        Stamps8 tsb;
        const TileRec8* TT = (const TileRec8*)(sm + S.tt);
        const int ib = ia < T ? ia : 0;
        const long long ob = TT[ib].off;
        const uint8_t* baseB = a.job[0].w.rows + (((long long)uniform_int((int)(ob >> 32)) << 32) | (long long)(unsigned)uniform_int((int)(ob & 0xffffffffll)));
        const int rbB = na >= 2 ? a.job[2].w.row_bytes : (na == 1 ? a.job[1].w.row_bytes : a.job[0].w.row_bytes);
        v8_phase<TB>(a, sm, S, C, T, baseB, uniform_int(TT[ib].nvalid), rbB, lane, wv, []() {}, tsb, false);
    } else {
        v8_phase<TA>(a, sm, S, C, T, base0, nvalid0, rbA, lane, wv, hook, ts, trace);
    }
    const unsigned long long t3 = trace ? clock64_dev() : 0ull;
    __syncthreads();
    // ---- epilogue pass ----
    const float* RES = (const float*)(sm + S.res);
    const float res = RES[tid & (kV8MaxTiles * 16 - 1)];
    if (a.gateup) {   // rows 0..7 of the tile = gate, 8..15 = up
        const float up = RES[(tid & (kV8MaxTiles * 16 - 1)) ^ 8];
        if (e_own) a.out[e_r] = f16_bits_to_f32(a.silu_tab[f32_to_f16_bits(res)]) * up;
    } else if (e_own) {
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
        } else {   // EPI_ROPE_Q / EPI_ROPE_K: rows (2i, 2i + 1) are one rotation (reference ggml.c:12536-12537, fma forms of the build)
            const float other = RES[tid ^ 1];
            const float o = (e_r & 1) ? fmaf(res, e_cs.x, other * e_cs.y) : fmaf(res, e_cs.x, -(other * e_cs.y));
            if (e_epi == EPI_ROPE_Q) a.q_f16[e_r] = f32_to_f16_bits(o);
            else a.kcache[kcache_off(pos, e_r, a.head_dim, a.n_ctx)] = f32_to_f16_bits(o);
        }
    }
    if (trace) { tr[0] = t0; tr[2] = t2; tr[3] = t3; tr[6] = clock64_dev(); tr[8] = ts.first_math; tr[9] = ts.first_arrive; tr[10] = ts.loop_end; tr[11] = ts.duty_end; }
}
