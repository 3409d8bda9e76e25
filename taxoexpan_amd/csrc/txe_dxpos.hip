// First GATLayer of PGAT, backward of `fc(cat(h, Emb[pos]))` (model_zoo.py:82-83, :214-215) with respect to the POSITION columns
// only -- the raw features h carry no gradient, so of d_X = d_Y W just the <= 64 columns behind the position embedding are needed,
// and those only as per-class sums (the embedding gradient dP[c][j] = sum_{pos[m] == c} d_X[m][Kh + j]).
//
// As a GEMM this product is N x 52 with K = 2,048: 140 row panels for 256 CUs, latency-bound (0.10 of the MFMA roof, 225 us on the
// 18 k-node training batch even on a second stream).  It is really ONE PASS OVER d_Y (144 MB) with 104 flops per element:
//
//   * one workgroup per 16 rows of d_Y (1,118 workgroups of the training batch: all resident at once, 4.4 per CU), its four waves
//     take a quarter of the K = Fp reduction each and meet in LDS -- no split-K partials in HBM, no fix-up launch;
//   * v_mfma_f32_16x16x4_f32 with BOTH operands fetched straight from global memory into the MFMA's lane layout, no LDS staging:
//     the MFMA sums over k in any order as long as A and B agree, so lane (row i, k-quarter q) loads the 16 bytes
//     d_Y[i][k + 4q .. 4q+3] and feeds element t to MFMA t (k = k + 4q + t);  the B side reads Wp[k + 4q + t][c0 + 4n .. 4n+3] --
//     16 bytes along the OUTPUT columns -- and hands element e to the accumulator of "column block e" = columns {4n + e}: a
//     permutation of the 64 output columns that the epilogue undoes for free (a lane then owns 4 consecutive columns of a row).
//     W's position slab (Fp x 52 floats = 426 KB) stays in every XCD's L2;
//   * all loads of a chunk (2 x 16 k) are issued before the chunk in flight is multiplied (two register sets), ~5 waves per SIMD;
//   * epilogue in the same launch: sum of the four k-quarters in fixed order, dropout keep bits, d_X store (3.7 MB), and the
//     workgroup's per-class partial sums for dP -- the `pos_segsum_stage1` pass over d_X disappears.
#include "txe_dxpos.h"

#include <stdlib.h>

namespace txe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DX_LD = 68;            // row stride of the k-quarter tiles in LDS (floats)

// CS: 16-k steps per chunk.  PF: the NEXT chunk's d_Y vectors (the HBM stream) are requested before the current chunk is multiplied;
// the weight vectors (L2 hits) are fetched at the head of their own chunk, FIRST, so that waiting for them (the in-order vmcnt
// counter) never waits for the prefetch behind them.  The other ~4 waves of the SIMD cover what latency is left.
template <int CS, bool PF, int MINW>
__global__ __launch_bounds__(256, MINW) void gat_dx_pos_kernel(const DxPosArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][DXPOS_ROWS][DX_LD];
    __shared__ int s_pos[DXPOS_ROWS];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int i = l & 15, kq = l >> 4;
    const int r0 = blockIdx.x * DXPOS_ROWS;
    if (threadIdx.x < DXPOS_ROWS) s_pos[threadIdx.x] = (r0 + (int)threadIdx.x < a.n_rows) ? a.pos[r0 + threadIdx.x] : -1;

    const int kw = a.K >> 2;                                       // this wave's k range [w*kw, (w+1)*kw), a multiple of 32
    const int row = min(r0 + i, a.n_rows - 1);                     // rows past the end re-read the last one (never stored)
    const float* ap = a.dY + (long long)row * a.ld_dy + w * kw + 4 * kq;
    const int cofs = min(a.c0 + 4 * i, a.Kp - 4);                  // column vectors past the matrix re-read its last one (never stored)
    const int ldw = (int)a.ld_w;
    const float* bu = a.Wp + (long long)(w * kw) * ldw;            // wave-uniform part of the weight addresses
    const unsigned boff = (unsigned)(4 * kq * ldw + cofs);         // this lane's part

    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};

#define TXE_DX_LOAD_A(av_)                                                                                           \
    {                                                                                                                \
        _Pragma("unroll") for (int s = 0; s < CS; ++s) av_[s] = *reinterpret_cast<const float4*>(ap + 16 * s);        \
        ap += 16 * CS;                                                                                               \
    }
#define TXE_DX_LOAD_B(bv_)                                                                                           \
    {                                                                                                                \
        _Pragma("unroll") for (int s = 0; s < CS; ++s)                                                               \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                            \
                bv_[s][t] = *reinterpret_cast<const float4*>(bu + (16 * s + t) * ldw + boff);                        \
        bu += 16 * CS * ldw;                                                                                         \
    }
#define TXE_DX_MFMA(av_, bv_, s_, t_, e_, AE_, BE_) \
    acc[e_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[s_].AE_, bv_[s_][t_].BE_, acc[e_], 0, 0, 0);
#define TXE_DX_STEP_T(av_, bv_, s_, t_, AE_)                                                                         \
    TXE_DX_MFMA(av_, bv_, s_, t_, 0, AE_, x) TXE_DX_MFMA(av_, bv_, s_, t_, 1, AE_, y) TXE_DX_MFMA(av_, bv_, s_, t_, 2, AE_, z) \
    TXE_DX_MFMA(av_, bv_, s_, t_, 3, AE_, w)
#define TXE_DX_COMPUTE(av_, bv_)                                                                                     \
    {                                                                                                                \
        _Pragma("unroll") for (int s = 0; s < CS; ++s) {                                                             \
            TXE_DX_STEP_T(av_, bv_, s, 0, x) TXE_DX_STEP_T(av_, bv_, s, 1, y) TXE_DX_STEP_T(av_, bv_, s, 2, z)       \
            TXE_DX_STEP_T(av_, bv_, s, 3, w)                                                                         \
        }                                                                                                            \
    }

    const int nchunk = kw / (16 * CS);
    if constexpr (!PF) {
        for (int c = 0; c < nchunk; ++c) {
            float4 av[CS], bv[CS][4];
            TXE_DX_LOAD_B(bv)
            TXE_DX_LOAD_A(av)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_COMPUTE(av, bv)
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        float4 a0[CS], a1[CS], bv[CS][4];
        TXE_DX_LOAD_A(a0)
        int c = 0;
        for (; c + 2 < nchunk; c += 2) {                           // a0 holds chunk c
            TXE_DX_LOAD_B(bv)
            __builtin_amdgcn_sched_barrier(0);                     // (weights first: see above)
            TXE_DX_LOAD_A(a1)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_COMPUTE(a0, bv)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_LOAD_B(bv)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_LOAD_A(a0)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_COMPUTE(a1, bv)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < nchunk) {                                      // the last one or two chunks: nothing is fetched past the row
            TXE_DX_LOAD_B(bv)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_LOAD_A(a1)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_COMPUTE(a0, bv)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_LOAD_B(bv)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_COMPUTE(a1, bv)
        } else {
            TXE_DX_LOAD_B(bv)
            __builtin_amdgcn_sched_barrier(0);
            TXE_DX_COMPUTE(a0, bv)
        }
    }
#undef TXE_DX_COMPUTE
#undef TXE_DX_STEP_T
#undef TXE_DX_MFMA
#undef TXE_DX_LOAD_A
#undef TXE_DX_LOAD_B

    // accumulator register r of column block e: row 4*(lane >> 4) + r, output column 4*(lane & 15) + e
#pragma unroll
    for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(&red[w][4 * kq + r][4 * i]) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    __syncthreads();

    // every thread finishes 4 consecutive columns of one row: k-quarters in fixed order, keep bits, store
    const int rr = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int m = r0 + rr, mc = min(m, a.n_rows - 1);
    const int gc = a.c0 + 4 * cq;                                  // (4 consecutive columns from a multiple of 4 share a mask word)
    const unsigned mwd = a.mask[a.mask_on ? ((long long)mc * a.mask_ld + min((long long)(gc >> 5), a.mask_ld - 1)) : 0];
    float4 v = *reinterpret_cast<const float4*>(&red[0][rr][4 * cq]);
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) {
        const float4 u = *reinterpret_cast<const float4*>(&red[ww][rr][4 * cq]);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const unsigned kb = a.mask_on ? (mwd >> (gc & 31)) : 0xFu;
    v.x = (kb & 1u) ? v.x * a.drop_scale : 0.f;
    v.y = (kb & 2u) ? v.y * a.drop_scale : 0.f;
    v.z = (kb & 4u) ? v.z * a.drop_scale : 0.f;
    v.w = (kb & 8u) ? v.w * a.drop_scale : 0.f;
    const bool cok = 4 * cq < a.NC;
    if (a.dX != nullptr && m < a.n_rows && cok) *reinterpret_cast<float4*>(a.dX + (long long)m * a.ld_dx + gc) = v;
    *reinterpret_cast<float4*>(&red[0][rr][4 * cq]) = v;           // (this thread's own four words of the first tile)
    __syncthreads();

    // per-class partial sums of the position columns over this workgroup's rows
    for (int t = threadIdx.x; t < a.vocab * a.Pd; t += 256) {
        const int cls = t / a.Pd, j = t - cls * a.Pd;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < DXPOS_ROWS; ++r) s += (s_pos[r] == cls) ? red[0][r][a.pcol0 + j] : 0.f;
        a.ppart[((long long)blockIdx.x * a.vocab + cls) * a.Pd + j] = s;
    }
}

int dxpos_launch(const DxPosArgs& a_in, hipStream_t stream) {
    DxPosArgs a = a_in;
    if (a.n_rows <= 0) return TXE_OK;
    if (!a.dY || !a.Wp || !a.pos || !a.ppart || (a.K & 127) || a.NC < 1 || a.NC > DXPOS_MAXC || (a.c0 & 3) || (a.Kp & 3) || a.Kp < 4 ||
        (a.ld_dy & 3) || (a.ld_w & 3) || (a.ld_dx & 3) || a.pcol0 < 0 || a.pcol0 + a.Pd > DXPOS_MAXC || a.vocab < 1 || a.Pd < 1 ||
        (long long)a.K * a.ld_w > 0x1fffffffll)
        return TXE_ERR_ARG;
    if (!a.mask_on) { a.mask = reinterpret_cast<const unsigned*>(a.Wp); a.mask_ld = 1; a.drop_scale = 1.f; }
    // algorithmic bytes: d_Y once, the weight slab once, the outputs once
    ProfScope prof("gat_dx_pos_kernel", stream, 4.0 * ((double)a.n_rows * a.K + (double)a.K * a.NC + (double)a.n_rows * a.NC), 1);
    static int variant = -1;                                      // tuning switch: TXE_DXPOS_VARIANT = 0 (2 steps, no prefetch) | 1 (1 step,
    if (variant < 0) { const char* e = getenv("TXE_DXPOS_VARIANT"); variant = e ? atoi(e) : 1; }        // prefetch) | 2 (2 steps, prefetch)
    const dim3 grid(dxpos_blocks(a.n_rows));
    if (variant == 0) hipLaunchKernelGGL((gat_dx_pos_kernel<2, false, 5>), grid, dim3(256), 0, stream, a);
    else if (variant == 1) hipLaunchKernelGGL((gat_dx_pos_kernel<1, true, 5>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gat_dx_pos_kernel<2, true, 4>), grid, dim3(256), 0, stream, a);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // namespace txe
