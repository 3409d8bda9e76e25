// fp32 products on the bf16 matrix pipe: C = A B^T with every fp32 operand element carried as the EXACT sum of three bf16 numbers
// (x = x1 + x2 + x3: 8 + 8 + 8 significand bits) and six of the nine plane products formed by v_mfma_f32_32x32x16_bf16 with fp32
// accumulation -- a1b1, a1b2, a2b1, a1b3, a2b2, a3b1.  Each plane product is exact in fp32 (8 x 8 bits); the three dropped terms are
// below 2^-26 |a b|, i.e. below the rounding of ONE fp32 multiply, and an element sees 6 K/16 accumulator roundings instead of the
// fp32 MFMA's K/2 -- the result is as close to the exact product as the fp32 path's (tests: both against float64).  The bf16 pipe
// runs 16x the fp32 MFMA rate (MI355X: 2.5 PFLOP/s dense against 157.3 TFLOP/s), so six products per k cost 6/16 of the fp32 time.
//
// Packed operand ("planes"): a matrix of R rows x K columns is stored as 1-KB MFMA fragments,
//     fragment (rb, kt, p) = rows-slots [32 rb, 32 rb + 32) x columns [16 kt, 16 kt + 16) of plane p, at byte ((rb * nkt + kt) * 3 + p) * 1024,
//     inside it lane l = (column half kh = l >> 5, slot s = l & 31) owns the 16 bytes at l * 16: columns 16 kt + 8 kh .. + 7 of slot s
// -- exactly what ONE global_load_lds_dwordx4 of a wave copies into LDS and what ONE ds_read_b128 hands the MFMA as its A / B operand.
// Slot -> row: side 0 (the A operand, C's rows): row = 32 rb + s.  Side 1 (the B operand, C's columns): a 128-column tile is four
// fragments j = rb & 3 and slot s of fragment j is column 128 (rb >> 2) + 64 (j >> 1) + 2 s + (j & 1), so that a lane's two
// accumulator blocks hold two ADJACENT columns of C (8-byte stores, 256 contiguous bytes per half wave).
// Rows are padded to 256 and columns to 16 with zeros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace txe {

constexpr int SPL_BM = 128, SPL_BN = 128, SPL_KT = 16;
constexpr int SPL_FRAG_BYTES = 1024;

static inline size_t split_packed_bytes(int rows, int cols) {
    const size_t rb = (size_t)((rows + 255) / 256) * 8, nkt = (size_t)(cols + SPL_KT - 1) / SPL_KT;
    return rb * nkt * 3 * SPL_FRAG_BYTES;
}

// x -> (x1, x2, x3) as bf16 bit patterns, round-to-nearest-even at each level (x - x1 and (x - x1) - x2 are exact in fp32)
__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = bf16_rne(x);
    const float r1 = x - __uint_as_float(p1 << 16);
    p2 = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(p2 << 16);
    p3 = bf16_rne(r2);
}
// eight consecutive columns of one row -> the three planes' 16-byte lane words
__device__ __forceinline__ void split3x8(const float (&x)[8], uint4& w1, uint4& w2, uint4& w3) {
    unsigned a[8], b[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split3(x[i], a[i], b[i], c[i]);
    w1 = make_uint4(a[0] | (a[1] << 16), a[2] | (a[3] << 16), a[4] | (a[5] << 16), a[6] | (a[7] << 16));
    w2 = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
    w3 = make_uint4(c[0] | (c[1] << 16), c[2] | (c[3] << 16), c[4] | (c[5] << 16), c[6] | (c[7] << 16));
}
// row of slot s of fragment block rb
__device__ __host__ __forceinline__ int split_slot_row(int side, int rb, int s) {
    return side == 0 ? 32 * rb + s : 128 * (rb >> 2) + 64 * ((rb & 3) >> 1) + 2 * s + (rb & 1);
}

// launches (txe_gemm_split.hip)
int split_pack_launch(const float* src, long long ld, int rows, int cols, int side, void* packed, hipStream_t stream);
int gemm_nt_split_launch(const void* Ap, const void* Bp, int M, int N, int K, float* C, long long ldc, double alg_flops, hipStream_t stream);

}  // namespace txe
