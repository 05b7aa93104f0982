// GGML block formats as stored in model files, and the GPU-side "plane" layout they are repacked into at load time.
//
// File layouts (reference models/ggml/ggml.c:888-925 Q4_0/Q8_0; models/ggml/k_quants.h:76-126 Q4_K/Q5_K/Q6_K):
//   Q4_0  18 B / 32 w : f16 d | 16 B nibbles (lo nibble = w[j], hi nibble = w[j+16])
//   Q8_0  34 B / 32 w : f16 d | 32 x int8
//   Q4_K 144 B / 256 w: f16 d | f16 dmin | 12 B 6-bit scales+mins | 128 B nibbles
//   Q5_K 176 B / 256 w: f16 d | f16 dmin | 12 B scales | 32 B high bits | 128 B nibbles
//   Q6_K 210 B / 256 w: 128 B low nibbles | 64 B high 2-bits | 16 x int8 scales | f16 d
//
// Repacked planes (one-time, values unchanged): the byte streams a wavefront reads with 16-byte loads are made
// contiguous per row and 16-byte aligned (a Q6_K row of K=11008 is 9030 B in the file: rows are only 2-byte aligned):
//   Q4_K: p0 = qs[M][nb*128]  p1 = hdr[M][nb*16] (d,dmin,scales = the first 16 file bytes)
//   Q5_K: p0 = qs[M][nb*128]  p1 = hdr[M][nb*16]  p2 = qh[M][nb*32]
//   Q6_K: p0 = ql[M][nb*128]  p1 = sc[M][nb*16]   p2 = qh[M][nb*64]  p3 = d[M][nb] (f16)
//   Q8_0: p0 = qs[M][K]                                              p3 = d[M][K/32] (f16)
//   Q4_0: p0 = qs[M][K/2]                                            p3 = d[M][K/32] (f16)
// Total bytes are identical to the file (no padding inside planes), so the roofline byte count is unchanged.
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
    switch (t) { case GT_F32: case GT_F16: return 1; case GT_Q4_0: case GT_Q8_0: return 32;
                 case GT_Q4_K: case GT_Q5_K: case GT_Q6_K: return 256; default: return 0; }
}
CT_HD static inline int ggml_block_bytes(int t) {
    switch (t) { case GT_F32: return 4; case GT_F16: return 2; case GT_Q4_0: return 18; case GT_Q8_0: return 34;
                 case GT_Q4_K: return 144; case GT_Q5_K: return 176; case GT_Q6_K: return 210; default: return 0; }
}
CT_HD static inline size_t ggml_row_bytes(int t, int64_t k) { return (size_t)(k / ggml_block_elems(t)) * ggml_block_bytes(t); }
CT_HD static inline bool is_kquant(int t) { return t == GT_Q4_K || t == GT_Q5_K || t == GT_Q6_K; }

// A weight matrix resident on one GPU.  M rows (outputs), K columns (inputs).
struct DevMat {
    int type = -1;
    int M = 0, K = 0;
    int nb = 0;                 // blocks per row (K/256 for K-quants, K/32 for Q4_0/Q8_0)
    const uint8_t* p[4] = {nullptr, nullptr, nullptr, nullptr};
    const uint8_t* raw = nullptr;  // file layout, kept only for tensors used by row lookup (token_embd)
    size_t bytes = 0;           // total device bytes (== file bytes)
};
