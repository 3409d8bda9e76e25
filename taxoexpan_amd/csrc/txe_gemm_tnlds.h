// Split-K weight-gradient products C[z] = A^T B (A [K][M], B [K][N], both row-contiguous and PLAIN) on 128 x 160 tiles with the
// operand tiles copied global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write, no "finish" phase).
//
// A row-contiguous operand tile [32 k][rows] has the SAME layout in LDS as in HBM, so a wave's 64 lanes x 16 bytes are exactly 1 KB
// of consecutive LDS words -- what the LDS-direct load writes (M0 base + lane * 16).  The generic kernel's loop spent a fifth of its
// issue slots on the round trip through VGPRs (txe_gemm.h ablation: 125.7 -> 141 TF/s without its global loads, 147 without the
// stage stores); here the k-loop is: 9 async copies for tile t+1, the MFMA block of tile t, s_waitcnt vmcnt(0), barrier.
// B fragments use a column permutation instead of 20 scalar reads per k-step: lane n owns output columns 4n..4n+3 (blocks 0-3: one
// ds_read_b128 per k) and column 128+n (block 4: one ds_read_b32); the epilogue then stores 16-byte vectors straight from the
// accumulators.  Accumulation order per element = the k order of gemm_kernel (one MFMA per 2 k, k-groups of 8): bit-identical partials.
#pragma once
#include "txe_gemm.h"

namespace txe {

struct TnLds {
    const float* A; long long lda; int M;
    const float* B; long long ldb; int N;
    int K, ksplit;
    float* C; long long ldc; long long split_stride;
};

constexpr int TL_BM = 128, TL_BN = 160, TL_BK = 32;
constexpr int TL_ASZ = TL_BK * TL_BM, TL_BSZ = TL_BK * TL_BN;      // floats per stage

__global__ __launch_bounds__(256, 2) void gemm_tn_lds_kernel(const TnLds p) {
    // one LDS object per stage: the compiler waits (vmcnt) for an LDS-direct copy before any ds_read it cannot prove independent of it --
    // with the stages as distinct objects and the loop unrolled by two, the copy into one stage is provably not what the MFMA block
    // reads from the other
    __shared__ __attribute__((aligned(16))) float st0[TL_ASZ + TL_BSZ];
    __shared__ __attribute__((aligned(16))) float st1[TL_ASZ + TL_BSZ];
    const int ntiles = gridDim.x;
    const int vb = xcd_remap(blockIdx.x + ntiles * blockIdx.y, ntiles * gridDim.y);
    const int zslice = vb / ntiles, lb = vb % ntiles;
    const int nbn = (p.N + TL_BN - 1) / TL_BN;
    const int m0 = (lb / nbn) * TL_BM, n0 = (lb % nbn) * TL_BN;
    const int kbeg = zslice * p.ksplit, kend = min(p.K, kbeg + p.ksplit);
    const int nk = (kend - kbeg + TL_BK - 1) / TL_BK;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;

    // this lane's source addresses of tile 0 (clamped columns: vectors past M / N re-read the last one inside the row pitch -- they
    // only feed rows / columns of C past M / N, which are never stored)
    const float* ap[4];
    const float* bp[5];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = w * 4 + q;
        const int krow = 2 * j + (l >> 5), col = min(m0 + (l & 31) * 4, (int)p.lda - 4);
        ap[q] = p.A + (long long)(kbeg + krow) * p.lda + col;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int e = (w * 5 + q) * 64 + l;
        const int krow = e / 40, col = min(n0 + (e % 40) * 4, (int)p.ldb - 4);
        bp[q] = p.B + (long long)(kbeg + krow) * p.ldb + col;
    }
    const long long adv_a = (long long)TL_BK * p.lda, adv_b = (long long)TL_BK * p.ldb;

    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    typedef __attribute__((address_space(3))) float lds_f;

#define TXE_TL_ISSUE(st_)                                                                                             \
    {                                                                                                                \
        lds_f* as = (lds_f*)(st_) + w * 4 * 256;                                                                     \
        lds_f* bs = (lds_f*)(st_) + TL_ASZ + w * 5 * 256;                                                            \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
            __builtin_amdgcn_global_load_lds(ap[q], as + q * 256, 16, 0, 0);                                         \
            ap[q] += adv_a;                                                                                          \
        }                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 5; ++q) {                                                              \
            __builtin_amdgcn_global_load_lds(bp[q], bs + q * 256, 16, 0, 0);                                         \
            bp[q] += adv_b;                                                                                          \
        }                                                                                                            \
    }
    /* the ragged last k-tile of a slice: rows at or past kend re-read row kend-1 (A's are zeroed afterwards) */
#define TXE_TL_ISSUE_LAST(st_, t_)                                                                                    \
    {                                                                                                                \
        lds_f* as = (lds_f*)(st_) + w * 4 * 256;                                                                     \
        lds_f* bs = (lds_f*)(st_) + TL_ASZ + w * 5 * 256;                                                            \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
            const int krow = kbeg + (t_) * TL_BK + 2 * (w * 4 + q) + (l >> 5);                                       \
            __builtin_amdgcn_global_load_lds(ap[q] - (long long)max(0, krow - (kend - 1)) * p.lda, as + q * 256, 16, 0, 0); \
        }                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 5; ++q) {                                                              \
            const int krow = kbeg + (t_) * TL_BK + ((w * 5 + q) * 64 + l) / 40;                                      \
            __builtin_amdgcn_global_load_lds(bp[q] - (long long)max(0, krow - (kend - 1)) * p.ldb, bs + q * 256, 16, 0, 0); \
        }                                                                                                            \
    }
#define TXE_TL_ZERO_TAIL(st_, t_)                                                                                     \
    {                                                                                                                \
        const int valid = kend - (kbeg + (t_) * TL_BK);                                                              \
        for (int i = valid * TL_BM + threadIdx.x; i < TL_ASZ; i += 256) (st_)[i] = 0.f;                              \
    }
#define TXE_TL_COMPUTE(st_)                                                                                           \
    {                                                                                                                \
        const float* a_l = (st_) + w * 32 + (l & 31);                                                                \
        const float* b4 = (st_) + TL_ASZ + 4 * (l & 31);                                                             \
        const float* b1 = (st_) + TL_ASZ + 128 + (l & 31);                                                           \
        _Pragma("unroll") for (int kb = 0; kb < TL_BK / 8; ++kb) {                                                   \
            const int kk = kb * 8 + (l >> 5) * 4;                                                                    \
            float fa[4], f1[4];                                                                                      \
            float4 f4[4];                                                                                            \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                          \
                fa[s] = a_l[(kk + s) * TL_BM];                                                                       \
                f4[s] = *reinterpret_cast<const float4*>(b4 + (kk + s) * TL_BN);                                     \
                f1[s] = b1[(kk + s) * TL_BN];                                                                        \
            }                                                                                                        \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                          \
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], f4[s].x, acc[0], 0, 0, 0);                      \
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], f4[s].y, acc[1], 0, 0, 0);                      \
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], f4[s].z, acc[2], 0, 0, 0);                      \
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], f4[s].w, acc[3], 0, 0, 0);                      \
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], f1[s], acc[4], 0, 0, 0);                        \
            }                                                                                                        \
        }                                                                                                            \
    }
#define TXE_TL_SYNC()                                                                                                 \
    {                                                                                                                \
        __builtin_amdgcn_s_waitcnt(0); /* vmcnt(0) expcnt(0) lgkmcnt(0): this wave's copies have landed */           \
        __syncthreads();                                                                                             \
    }

    const int nfull = (kend - kbeg) / TL_BK;          // whole k-tiles; nk == nfull + 1 when the slice ends in a ragged one
    // tile t lives in stage t & 1; tile t+1 is copied while tile t is multiplied (loop unrolled by two: static stage objects)
    if (nk > 0) {
        if (nfull > 0) TXE_TL_ISSUE(st0) else TXE_TL_ISSUE_LAST(st0, 0)
        TXE_TL_SYNC()
        if (nfull == 0) { TXE_TL_ZERO_TAIL(st0, 0) __syncthreads(); }
        int t = 1;
        for (; t + 1 < nfull; t += 2) {
            TXE_TL_ISSUE(st1)
            __builtin_amdgcn_sched_barrier(0);
            TXE_TL_COMPUTE(st0)
            __builtin_amdgcn_sched_barrier(0);
            TXE_TL_SYNC()
            TXE_TL_ISSUE(st0)
            __builtin_amdgcn_sched_barrier(0);
            TXE_TL_COMPUTE(st1)
            __builtin_amdgcn_sched_barrier(0);
            TXE_TL_SYNC()
        }
        // tiles t .. nk-1 remain (t odd: tile t-1 sits in st0); at most one more whole tile, then possibly the ragged one
        bool in0 = true;                              // the stage holding tile t-1
        if (t < nfull) {
            TXE_TL_ISSUE(st1)
            __builtin_amdgcn_sched_barrier(0);
            TXE_TL_COMPUTE(st0)
            __builtin_amdgcn_sched_barrier(0);
            TXE_TL_SYNC()
            ++t;
            in0 = false;
        }
        if (t < nk) {                                // the ragged tile (if it is not tile 0)
            if (in0) {
                TXE_TL_ISSUE_LAST(st1, t)
                __builtin_amdgcn_sched_barrier(0);
                TXE_TL_COMPUTE(st0)
                __builtin_amdgcn_sched_barrier(0);
                TXE_TL_SYNC()
                TXE_TL_ZERO_TAIL(st1, t)
            } else {
                TXE_TL_ISSUE_LAST(st0, t)
                __builtin_amdgcn_sched_barrier(0);
                TXE_TL_COMPUTE(st1)
                __builtin_amdgcn_sched_barrier(0);
                TXE_TL_SYNC()
                TXE_TL_ZERO_TAIL(st0, t)
            }
            __syncthreads();
            in0 = !in0;
        }
        if (in0) TXE_TL_COMPUTE(st0) else TXE_TL_COMPUTE(st1)
    }
#undef TXE_TL_SYNC
#undef TXE_TL_COMPUTE
#undef TXE_TL_ZERO_TAIL
#undef TXE_TL_ISSUE_LAST
#undef TXE_TL_ISSUE

    // accumulator register e of a 32 x 32 block: row (e&3) + 8*(e>>2) + 4*(lane>>5), block column lane&31 -> C columns 4*(lane&31) + j
    // (blocks 0-3) / 128 + (lane&31) (block 4)
    float* cb = p.C + (long long)zslice * p.split_stride;
    const int nc4 = n0 + 4 * (l & 31), nc1 = n0 + 128 + (l & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m0 + w * 32 + 4 * (l >> 5) + (e & 3) + 8 * (e >> 2);
        if (m < p.M) {
            if (nc4 + 3 < p.N) *reinterpret_cast<float4*>(cb + (long long)m * p.ldc + nc4) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
            else {
                if (nc4 < p.N) cb[(long long)m * p.ldc + nc4] = acc[0][e];
                if (nc4 + 1 < p.N) cb[(long long)m * p.ldc + nc4 + 1] = acc[1][e];
                if (nc4 + 2 < p.N) cb[(long long)m * p.ldc + nc4 + 2] = acc[2][e];
            }
            if (nc1 < p.N) cb[(long long)m * p.ldc + nc1] = acc[4][e];
        }
    }
}

}  // namespace txe
