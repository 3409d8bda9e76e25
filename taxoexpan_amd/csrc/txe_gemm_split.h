// fp32 products on the bf16 matrix pipe: C = A B^T with every fp32 operand element carried as the EXACT sum of three bf16 numbers
// (x = x1 + x2 + x3: 8 + 8 + 8 significand bits) and six of the nine plane products formed by v_mfma_f32_32x32x16_bf16 with fp32
// accumulation -- a1b1, a1b2, a2b1, a1b3, a2b2, a3b1.  Each plane product is exact in fp32 (8 x 8 bits); the three dropped terms are
// 2^-27.4 |a b| in the root mean square and at most 2^-23 |a b| (an fp32 multiply's own rounding: 2^-25.2 rms, at most 2^-24;
// tests/test_split_arithmetic.py), and an element sees 6 K/16 accumulator roundings instead of the fp32 MFMA's K/2 -- the result is as
// close to the exact product as the fp32 path's (tests/test_gpu_split_gemm.py: both against float64).  The bf16 pipe
// runs 16x the fp32 MFMA rate (MI355X: 2.5 PFLOP/s dense against 157.3 TFLOP/s), so six products per k cost 6/16 of the fp32 time.
//
// Packed operand ("planes"): a matrix of R rows x K columns is stored as 1-KB MFMA fragments,
//     fragment (rb, kt, p) = rows-slots [32 rb, 32 rb + 32) x columns [16 kt, 16 kt + 16) of plane p, at byte ((rb * nkt + kt) * 3 + p) * 1024,
//     inside it lane l = (column half kh = l >> 5, slot s = l & 31) owns the 16 bytes at l * 16: columns 16 kt + 8 kh .. + 7 of slot s
// -- exactly what ONE global_load_lds_dwordx4 of a wave copies into LDS and what ONE ds_read_b128 hands the MFMA as its A / B operand.
// (txe_split_pack sides 2 / 3: sides 0 / 1 of a matrix given as its transpose [K][ld >= R].)
// Slot -> row: side 0 (the A operand, C's rows): row = 32 rb + s.  Side 1 (the B operand, C's columns): a 128-column tile is four
// fragments j = rb & 3 and slot s of fragment j is column 128 (rb >> 2) + 64 (j >> 1) + 2 s + (j & 1), so that a lane's two
// accumulator blocks hold two ADJACENT columns of C (8-byte stores, 256 contiguous bytes per half wave).
// Rows are padded to 768 (whole tiles of 128, 192 or 256 rows) and columns to 16 with zeros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace txe {

constexpr int SPL_BM = 128, SPL_BN = 128, SPL_KT = 16;
constexpr int SPL_FRAG_BYTES = 1024;

static inline size_t split_packed_bytes(int rows, int cols) {
    const size_t rb = (size_t)((rows + 767) / 768) * 24, nkt = (size_t)(cols + SPL_KT - 1) / SPL_KT;
    return rb * nkt * 3 * SPL_FRAG_BYTES;
}

// both operands of an M x N x K product, each on a 256-byte boundary
static inline size_t split_pair_bytes(int M, int N, int K) {
    return (split_packed_bytes(M, K) + 255) / 256 * 256 + (split_packed_bytes(N, K) + 255) / 256 * 256;
}

// x -> (x1, x2, x3) as bf16 bit patterns, round-to-nearest-even at each level (v_cvt_pk_bf16_f32; x - x1 and (x - x1) - x2 are exact in
// fp32 whatever the rounding, so x1 + x2 + x3 == x exactly)
typedef float split_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 split_b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    const split_f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, split_b2));
}
__device__ __forceinline__ void split3x2(float x0, float x1, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(w1 << 16), r1 = x1 - __uint_as_float(w1 & 0xffff0000u);
    w2 = pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(w2 << 16), q1 = r1 - __uint_as_float(w2 & 0xffff0000u);
    w3 = pk_bf16(q0, q1);
}
// ---- the whole fp32 domain ---------------------------------------------------------------------------------------------------------
// The three-plane form is exact for |x| < 2^120 down to 2^-100 (every plane a normal bf16 number or zero).  Above that -- +-Inf, NaN,
// |x| >= 2^120 (x1 may round to Inf, and Inf - Inf poisons the residuals) -- an element is EXCEPTIONAL, and a packed fragment (32 slots x
// 16 contraction columns) that holds one is stored RAW instead: plane 1 = the high halves of the fp32 words, plane 2 = the low halves,
// plane 3 = 0x7FC0 (a bf16 NaN) in every element.  The NaN plane poisons every accumulator the fragment touches, the product kernels look
// at their accumulators once after the k-loop (anything not finite), and a tile that finds one recomputes itself with fp32 FMAs over
// operands decoded back to their exact fp32 values (split_decode8): IEEE results -- Inf stays Inf, Inf * 0 and Inf - Inf are NaN, a NaN
// stays in its row / column -- at scalar speed, for such tiles only.  (A tile whose exact result overflows recomputes itself too and
// arrives at the same Inf.)  The TN product's fp32 operand is split in its loader: Inf / NaN / beyond-bf16 elements poison its accumulators
// by themselves (Inf - Inf in the residual).
// Below 2^-100 the lower planes sink towards bf16's smallest numbers: a nonzero |x| < 2^-100 is carried with an ABSOLUTE error of at most
// 2^-126 (FLT_MIN) instead of fp32's relative 2^-24 -- what a flush-to-zero fp32 unit does to subnormals, extended to 2^-100.  Such
// elements are ordinary in gradients (softmax tails: alpha ~ e^-90), so they must not leave the fast path; their products are below
// 2^-100 |b| and matter to a result only when every other term of the dot product is that small too.
constexpr unsigned SPL_RAW_MARK = 0x7FC07FC0u;
__device__ __forceinline__ bool split_exceptional(float x) {
    return (__float_as_uint(x) & 0x7fffffffu) >= (247u << 23);      // |x| >= 2^120 (exponent field 247), Inf, NaN
}
__device__ __forceinline__ void split_raw2(float x0, float x1, unsigned& w1, unsigned& w2, unsigned& w3) {
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    w1 = (b0 >> 16) | (b1 & 0xffff0000u);
    w2 = (b0 & 0xffffu) | (b1 << 16);
    w3 = SPL_RAW_MARK;
}
// a lane word of each plane -> the eight fp32 values it stands for (exact: the planes' sum, or the raw halves of a marked fragment)
__device__ __forceinline__ void split_decode8(const uint4& w1, const uint4& w2, const uint4& w3, float (&x)[8]) {
    const unsigned a[4] = {w1.x, w1.y, w1.z, w1.w}, b[4] = {w2.x, w2.y, w2.z, w2.w}, c[4] = {w3.x, w3.y, w3.z, w3.w};
    const bool raw = c[0] == SPL_RAW_MARK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (raw) {
            x[2 * i] = __uint_as_float((a[i] << 16) | (b[i] & 0xffffu));
            x[2 * i + 1] = __uint_as_float((a[i] & 0xffff0000u) | (b[i] >> 16));
        } else {
            x[2 * i] = (__uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16)) + __uint_as_float(c[i] << 16);
            x[2 * i + 1] = (__uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(b[i] & 0xffff0000u)) + __uint_as_float(c[i] & 0xffff0000u);
        }
    }
}
__device__ __forceinline__ bool not_finite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }

// eight consecutive contraction elements of one row / column -> the three planes' 16-byte lane words
__device__ __forceinline__ void split3x8(const float (&x)[8], uint4& w1, uint4& w2, uint4& w3) {
    split3x2(x[0], x[1], w1.x, w2.x, w3.x);
    split3x2(x[2], x[3], w1.y, w2.y, w3.y);
    split3x2(x[4], x[5], w1.z, w2.z, w3.z);
    split3x2(x[6], x[7], w1.w, w2.w, w3.w);
}
// ... of a fragment that holds an exceptional element: raw halves + the NaN plane
__device__ __forceinline__ void split_raw8(const float (&x)[8], uint4& w1, uint4& w2, uint4& w3) {
    split_raw2(x[0], x[1], w1.x, w2.x, w3.x);
    split_raw2(x[2], x[3], w1.y, w2.y, w3.y);
    split_raw2(x[4], x[5], w1.z, w2.z, w3.z);
    split_raw2(x[6], x[7], w1.w, w2.w, w3.w);
}
__device__ __forceinline__ bool split_exceptional8(const float (&x)[8]) {
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 8; ++q) bad |= split_exceptional(x[q]);
    return bad;
}
__device__ __forceinline__ void split3x4(const float (&x)[4], uint2& w1, uint2& w2, uint2& w3) {
    split3x2(x[0], x[1], w1.x, w2.x, w3.x);
    split3x2(x[2], x[3], w1.y, w2.y, w3.y);
}
// row of slot s of fragment block rb
__device__ __host__ __forceinline__ int split_slot_row(int side, int rb, int s) {
    return side == 0 ? 32 * rb + s : 128 * (rb >> 2) + 64 * ((rb & 3) >> 1) + 2 * s + (rb & 1);
}

// ---- the TN product C = A^T B over the ROWS of A [n][M] and B [n][N] (weight gradients: A = d_Y, B = X, n = nodes) ------------------
// B comes packed CONTRACTION-major ("side 2", written once per step by split_pack_t_launch): fragment (nt, kb, p) = rows
// [16 nt, 16 nt + 16) x the 32 column slots of block kb, at byte ((nt * nkb + kb) * 3 + p) * 1024; lane (nh = l >> 5, s = l & 31) owns rows
// 16 nt + 8 nh .. + 7 of slot s.  Slot -> column inside a 160-column tile h (blocks kb = 5 h + j): j < 4: column 160 h + 4 s + j,
// j == 4: column 160 h + 128 + s -- a lane's five accumulator blocks are then four ADJACENT columns of C (one 16-byte store) and one more.
// Rows past the end and the columns that fill the last 160-column tile are zeros.  A is read as fp32 and split in the product's loader (each element once per column tile).
static inline size_t split_packed_t_bytes(int rows, int cols) {            // (whole 160-column tiles: the last one zero-filled)
    return (size_t)((rows + 15) / 16) * (size_t)(((cols + 159) / 160) * 5) * 3 * SPL_FRAG_BYTES;
}
static inline bool split_tn_eligible(int M, int N) { return M % 128 == 0 && N % 4 == 0 && M > 0 && N > 0; }
// ... and A [n_rows][lda] is addressed with 32-bit byte offsets (kept below 2^31: no reliance on how the instruction extends them)
static inline bool split_tn_fits(int n_rows, long long lda) { return (double)n_rows * (double)lda * 4.0 < 2147483648.0; }

// launches (txe_gemm_split.hip)
int split_pack_launch(const float* src, long long ld, int rows, int cols, int side, void* packed, hipStream_t stream);
// epilogue extras of the NT product (NULL = plain): C = acc * (mask bit of column + mask_col0 ? drop_scale : 0) * (act_src[m][c] > 0 or c >= cols_act
// ? 1 : act_slope) -- what txe_gemm.h's epi_store_one applies (mask == NULL / act_src == NULL: that factor is 1)
struct SplitEpi {
    const unsigned* mask; int mask_ld, mask_col0; float drop_scale;
    const float* act_src; long long ld_act; float act_slope; int cols_act;
};
int gemm_nt_split_launch(const void* Ap, const void* Bp, int M, int N, int K, float* C, long long ldc, double alg_flops, hipStream_t stream,
                         const SplitEpi* epi = nullptr);
int split_pack_t_launch(const float* src, long long ld, int rows, int cols, void* packed, hipStream_t stream);
struct Epi;
int gemm_nt_split_epi_launch(const void* Ap, const void* Bp, const Epi& E, int M, int N, int K, hipStream_t stream);
// X given by its sources instead of stored (split_pack_job): [h | Emb[pos]] with the feature dropout of mask (NULL: none) applied
struct SplitVSrc { const float* h; long long ld_h; const int* pos; const float* P; int Kh, Pd; const unsigned* mask; int wpr; float scale; };
int split_pack_layer_launch(const float* X, long long ldx, int n, const float* W, long long ldw, int f, int K, int Kt_cols, void* Xs, void* Ws,
                            void* Xt, hipStream_t stream, const SplitVSrc* vs = nullptr);
// part[z][M][ldc] = A[rows of slice z]^T B[rows of slice z], z < S, slices of ksplit rows (a multiple of 16)
int gemm_tn_split_launch(const float* A, long long lda, int M, const void* Bt, int N, int n_rows, int S, int ksplit, float* part, long long ldc,
                         long long split_stride, double alg_flops, hipStream_t stream);

}  // namespace txe
