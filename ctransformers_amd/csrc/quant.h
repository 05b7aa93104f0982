// GGML block formats as stored in model files, and the GPU-side layouts they are repacked into at load time
// (engine.cc:upload_matrix; values and total bytes unchanged, so the roofline byte count is the file's).
//
// File layouts (reference models/ggml/ggml.c:888-925 Q4_0/Q8_0; models/ggml/k_quants.h:76-126 Q4_K/Q5_K/Q6_K):
//   Q4_0  18 B / 32 w : f16 d | 16 B nibbles (lo nibble = w[j], hi nibble = w[j+16])
//   Q8_0  34 B / 32 w : f16 d | 32 x int8
//   Q4_K 144 B / 256 w: f16 d | f16 dmin | 12 B 6-bit scales+mins | 128 B nibbles
//   Q5_K 176 B / 256 w: f16 d | f16 dmin | 12 B scales | 32 B high bits | 128 B nibbles
//   Q6_K 210 B / 256 w: 128 B low nibbles | 64 B high 2-bits | 16 x int8 scales | f16 d
// (a Q6_K row of K=11008 is 9030 B in the file: rows are only 2-byte aligned — the repack is what makes 16-byte loads legal)
//
// LAYOUT_R2C4 (K-quants; kernels_pg.h prompt chunks): a record holds 2 rows x 4 consecutive K-blocks = 8 block
// slots (slot p = 4 * row + c), fields grouped so that a wavefront reads each field with 16-byte-per-lane loads; a row pair
// ("unit") = ceil(nb / 4) consecutive records, blocks past nb are zero slots.  The matrices of a launch site (attn_q | attn_k |
// attn_v; the fused gate/up matrix: unit u = gate row u, up row u) share one arena, so a launch walks one contiguous unit space.
//   Q4_K record 1152 B: hdr[8][16] | qs[8][128]
//   Q5_K record 1408 B: hdr[8][16] | qh[8][32] | qs[8][128]
//   Q6_K record 1680 B: d[8] f16 (16 B) | sc[8][16] | qh[8][64] | ql[8][128]
// Record sizes are 8 x the file block size (the repack moves bytes, engine.cc:place_kblock / repack_r2c4_kernel).
//
// LAYOUT_L9 (K-quants; kernels_v9.h decode mat-vec): the same records (2 rows x 4 consecutive K-blocks, same sizes, same unit
// spaces and arenas), bytes arranged per LANE of the wave that consumes a record: lane = 32 * row + 4 * l + c holds what AVX lane l
// of the reference touches in block slot (row, c) — so the lane that computes the integer lane sum sumi[l] also owns chain l.
//   Q4_K record 1152 B: qs[64 lanes][16] | hdr[8 slots][16]
//       lane dword j (0..3) = file qs bytes 32j + 4l .. +3 (low nibbles: vector 2j, high nibbles: vector 2j + 1, elements 4l .. 4l+3)
//       hdr = f16 d | f16 dmin | W1 | W2 | W3: the sixteen 6-bit scales / mins re-encoded (the file's 12 bytes, losslessly).  Every lane needs
//       all eight scales and ONE min (its own, m_l): the mins are 48 contiguous bits of the register pair W3:W2, so lane l takes
//       `(W3:W2 >> 6l) & 63` with one 64-bit shift and no word select; scale 7 pays for the spare-bit split once for all lanes:
//         W1 = sc0 | sc1<<6 | sc2<<12 | sc3<<18 | sc4<<24 | (sc7 & 3)<<30
//         W2 = m0 | m1<<6 | m2<<12 | m3<<18 | m4<<24 | (m5 & 3)<<30
//         W3 = (m5>>2) | m6<<4 | m7<<10 | sc5<<16 | sc6<<22 | (sc7>>2)<<28
//   Q5_K record 1408 B: qs[64][16] | qh[64][4] | hdr[8][16]      lane qh word = file qh bytes 4l .. 4l+3 (bit v = vector v)
//   Q6_K record 1680 B: ql[64][16] | qh[64][8] | sc[8][16] | d[8] f16
//       lane ql dwords (2n, 2n+1) = file ql bytes 64n + 4l.., 64n + 32 + 4l..; qh dword n = file qh bytes 32n + 4l..;
//       sc of a slot = the even scales (sc[0], sc[2], .. sc[14]) then the odd ones: lane l reads the 8 bytes of parity l >> 2
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__) && !defined(CT_EMU)
#define CT_HD __host__ __device__
#else
#define CT_HD
#endif

enum GgmlType : int {
    GT_F32 = 0, GT_F16 = 1, GT_Q4_0 = 2, GT_Q4_1 = 3, GT_Q5_0 = 6, GT_Q5_1 = 7, GT_Q8_0 = 8, GT_Q8_1 = 9,
    GT_Q2_K = 10, GT_Q3_K = 11, GT_Q4_K = 12, GT_Q5_K = 13, GT_Q6_K = 14, GT_Q8_K = 15,
};

CT_HD static inline int ggml_block_elems(int t) {
    switch (t) { case GT_F32: case GT_F16: return 1; case GT_Q4_0: case GT_Q8_0: case GT_Q4_1: case GT_Q5_0: case GT_Q5_1: return 32;
                 case GT_Q4_K: case GT_Q5_K: case GT_Q6_K: return 256; default: return 0; }
}
CT_HD static inline int ggml_block_bytes(int t) {
    switch (t) { case GT_F32: return 4; case GT_F16: return 2; case GT_Q4_0: return 18; case GT_Q8_0: return 34;
                 case GT_Q4_1: return 20; case GT_Q5_0: return 22; case GT_Q5_1: return 24;
                 case GT_Q4_K: return 144; case GT_Q5_K: return 176; case GT_Q6_K: return 210; default: return 0; }
}
CT_HD static inline size_t ggml_row_bytes(int t, int64_t k) { return (size_t)(k / ggml_block_elems(t)) * ggml_block_bytes(t); }
CT_HD static inline bool is_kquant(int t) { return t == GT_Q4_K || t == GT_Q5_K || t == GT_Q6_K; }
CT_HD static inline bool is_raw32(int t) { return t == GT_Q4_1 || t == GT_Q5_0 || t == GT_Q5_1; }   // kernels_raw32.h: file layout, token steps

// The 12-byte 6-bit scale/min field of Q4_K / Q5_K headers is re-encoded (losslessly, same size) as four 24-bit groups
// g_c = sc[2c] | sc[2c+1]<<6 | m[2c]<<12 | m[2c+1]<<18 (c = 0..3), little-endian bit order.
// LAYOUT_G4 (Q8_0 / Q4_0, kernels_q32.h): per 8-row tile and group of 4 consecutive 32-blocks one record with each lane's
// four dwords contiguous (Q8_0 1088 B, Q4_0 576 B = 8 rows x 4 blocks x the file block size: bytes unchanged).
// (LAYOUT_TILE8S = 2 was the 8-row tile layout of the retired mat-vec generations 5 / 6.)
enum { LAYOUT_TILE8S = 2, LAYOUT_G4 = 3, LAYOUT_R2C4 = 4, LAYOUT_L9 = 5, LAYOUT_F16 = 6, LAYOUT_RAW32 = 7, LAYOUT_M8 = 8 };   // LAYOUT_F16 / LAYOUT_RAW32: rows as in the file (DevMat::raw; kernels_f16.h, kernels_raw32.h)
CT_HD static inline int tile8_record_bytes(int t) { return 8 * ggml_block_bytes(t); }
// LAYOUT_R2C4 record of the prompt-chunk copy.  Q4_K / Q5_K: 8 x the file block.  Q6_K: the quants are stored UNPACKED-READY for the
// f16 matrix-core operand (kernels_pg.h): per (slot, p, l) two dwords A, B with the four 6-bit values of vector va(p) / vb(p),
// elements 4l .. 4l+3, at bits 3..8 | 19..24 (elements 0, 2) and 9..14 | 25..30 (elements 1, 3) — `(A & 0x01F801F8) | 0x58005800`
// and `((A >> 6) & 0x01F801F8) | 0x58005800` are the halves 128 + q6 of the operand, six instructions per AVX lane instead of
// twenty.  256 bytes per block instead of the file's 192 (ql + qh): d[8] | sc[8][16] | q[8 slots][4 p][8 l][A, B] = 2192 B.
CT_HD static inline int r2c4_record_bytes(int t) { return t == GT_Q6_K ? 2192 : tile8_record_bytes(t); }
// LAYOUT_L9 for the 32-element block types (kernels_v9.h): a record holds 2 rows x 16 consecutive blocks; lane (row, l, c) reads the
// dwords of blocks 4t + c (t = 0..3: the record's four chain sub-steps), elements 4l .. 4l+3, as ONE 16-byte load
//   Q8_0 record 1088 B: qs[64 lanes][4 t][4 B] | d[2 rows][4 c][4 t] f16          (2 x 16 x 34 B: the file's bytes)
//   Q4_0 record  576 B: qs[2 rows][4 (l & 3)][4 c][4 t][4 B] | d[2][4][4] f16      (lanes l and l + 4 share a dword: low / high nibbles)
CT_HD static inline int l9_record_bytes(int t) { return t == GT_Q8_0 ? 1088 : (t == GT_Q4_0 ? 576 : tile8_record_bytes(t)); }
// records per unit (row pair) of a row of K elements
CT_HD static inline int l9_spu(int t, int K) { return (t == GT_Q8_0 || t == GT_Q4_0) ? ((K >> 5) + 15) >> 4 : ((K >> 8) + 3) >> 2; }
CT_HD static inline bool is_block32(int t) { return t == GT_Q8_0 || t == GT_Q4_0; }

// LAYOUT_M8 (kernels_mm8.h: the order-free prompt kernels on the 32x32x32 int8 matrix cores): a record = one TILE of 32 rows x one K-STEP of 256
// elements, every field lane-linear for the wave that owns the tile — lane = 32 c + r reads, in one 16-byte load per piece, what row r contributes to
// K-chunk c (elements 16c .. 16c+15 of a 32-element sub-block) of the matrix instruction's B operand; a tile's records are consecutive along K.
// Record bytes = 32 x the file's bytes of 256 elements (the repack moves bytes; the 6-bit scales / mins of Q4_K / Q5_K are re-encoded, same 12 bytes).
//   Q4_K 4608 B: hdr[32 rows][16] | qs[4 g][64 lanes][16]                 piece g, lane (r, c) = file qs bytes 32g + 16c ..: low nibbles sub-block 2g, high 2g + 1
//       hdr = f16 d | f16 dmin | W1 | W2 | W3 with the sixteen 6-bit fields one bfe each (m7 in the three spare bit pairs):
//         W1 = sc0 | sc1<<6 | sc2<<12 | sc3<<18 | sc4<<24 | (m7 & 3)<<30
//         W2 = sc5 | sc6<<6 | sc7<<12 | m0<<18 | m1<<24 | ((m7>>2) & 3)<<30
//         W3 = m2 | m3<<6 | m4<<12 | m5<<18 | m6<<24 | (m7>>4)<<30
//   Q5_K 5632 B: hdr[32][16] | qh[64 lanes][16] | qs[4][64][16]           qh lane (r, c) = file qh bytes 16c ..: bit j of byte e = fifth bit of element 16c + e of sub-block j
//   Q6_K 6720 B: ql[2 h][2 o][64][16] | qh[2 h][64][16] | sc[32][16] | d[32] f16     ql piece (h, o), lane (r, c) = file ql bytes 64h + 32o + 16c ..; qh piece h = file qh bytes 32h + 16c ..
//   Q8_0 8704 B (8 blocks): d[32 rows][8] f16 | qs[8 j][64][16]           block j, lane (r, c) = its quants 16c .. 16c+15
//   Q4_0 4608 B (8 blocks): d[32][8] f16 | qs[8 j][32 rows][16]           both K-chunks of a row read the same 16 bytes (low nibbles: elements 0..15, high: 16..31)
// The fused gate/up matrix: tile T = gate rows 16T .. 16T+15 (r < 16) and up rows 16T .. 16T+15 (r >= 16).
CT_HD static inline int m8_record_bytes(int t) {
    switch (t) { case GT_Q4_K: return 4608; case GT_Q5_K: return 5632; case GT_Q6_K: return 6720; case GT_Q8_0: return 8704; case GT_Q4_0: return 4608; default: return 0; }
}
CT_HD static inline int m8_steps(int t, int K) { return (t == GT_Q8_0 || t == GT_Q4_0) ? ((K >> 5) + 7) >> 3 : K >> 8; }   // K-steps of 256 elements per row

// A weight matrix resident on one GPU.  M rows (outputs), K columns (inputs).
struct DevMat {
    int type = -1;
    int M = 0, K = 0;
    int nb = 0;                 // blocks per row (K/256 for K-quants, K/32 for Q4_0/Q8_0)
    const uint8_t* p[4] = {nullptr, nullptr, nullptr, nullptr};
    const uint8_t* raw = nullptr;  // file layout, kept only for tensors used by row lookup (token_embd)
    const uint8_t* r2 = nullptr;   // LAYOUT_R2C4 records (inside an arena several matrices may share): prompt-chunk kernels
    const uint8_t* r9 = nullptr;   // LAYOUT_L9 records, same arena geometry: decode mat-vec (kernels_v9.h)
    const uint8_t* m8 = nullptr;   // LAYOUT_M8 records (row tiles of 32): the order-free prompt kernels (kernels_mm8.h); null unless the handle runs them
    int layout = 0;                // LAYOUT_R2C4 (records in r2) / LAYOUT_G4 (records in p[0])
    size_t bytes = 0;           // total device bytes (== file bytes)
};
