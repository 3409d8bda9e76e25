// First GATLayer of PGAT, backward of `fc(cat(h, Emb[pos]))` (model_zoo.py:82-83, :214-215) with respect to the POSITION columns
// only -- the raw features h carry no gradient, so of d_X = d_Y W just the <= 64 columns behind the position embedding are needed,
// and those only as per-class sums (the embedding gradient dP[c][j] = sum_{pos[m] == c} d_X[m][Kh + j]).
//
// As a GEMM this product is N x 52 with K = 2,048: 140 row panels for 256 CUs, latency-bound (0.10 of the MFMA roof, 225 us on the
// 18 k-node training batch even on a second stream).  It is really ONE PASS OVER d_Y (144 MB) with 104 flops per element:
//
//   * the weight slab W[:, c0:c0+64] is STATIONARY: a workgroup (8 waves, one per CU) keeps one 512-row k-slice of it in LDS
//     (128 KB, loaded once) and streams its share of d_Y's rows past it; grid = k-slices x row groups = one workgroup per CU, every
//     CU gets the same number of 16-row blocks (+-1).  A first version without LDS (every wave fetching its weight fragments from
//     L2) moved 4x d_Y's bytes through the L1s and took 87 us;
//   * v_mfma_f32_16x16x4_f32, the A operand (d_Y) straight from global memory into the MFMA's lane layout: the MFMA sums over k in
//     any order as long as A and B agree, so lane (row i, k-quarter q) loads the 32 contiguous bytes d_Y[i][k + 8q .. 8q+7] of every
//     32-k step (the four lanes of a row cover one 128-byte line) and feeds element t of half h to the MFMA whose B operand is weight
//     row k + 8q + 4h + t;  B comes from LDS as 16 bytes along the OUTPUT columns, element e going to the accumulator of "column
//     block e" = columns {4n + e} -- a permutation of the 64 output columns that costs nothing to undo (a lane ends up owning 4
//     consecutive columns of a row: one 16-byte store);
//   * the 8 x 16-byte loads of the NEXT 128-k group are in flight while the current group's 128 MFMAs run (8 KB per wave, 64 KB per CU);
//   * the k-slices' raw partial products (KS x N x 64 floats: 18 MB against the 144 MB stream) are finished by `dxpos_finish_job`
//     (txe_dxpos.h) inside the layer's reduction launch: slices in fixed order, dropout keep bits, the d_X store and the per-class
//     partial sums of dP -- the `pos_segsum_stage1` pass over d_X disappears and no launch is added.
#include "txe_dxpos.h"
#include "txe_gemm_split.h"

#include <stdlib.h>

namespace txe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) short dx_bf16x8;

int device_cu_count();     // txe_profile.hip

// The product runs on the bf16 matrix pipe in fp32 accuracy (txe_gemm_split.h: every fp32 operand the exact sum of three bf16 planes, six
// plane products by v_mfma_f32_16x16x32_bf16, fp32 accumulation): 28 GF of plane products at the bf16 rate cost a third of what the
// 4.7 GF cost on the fp32 MFMA (v_mfma_f32_16x16x4_f32: 30 us at its peak, 48 us measured -- the kernel was compute-bound on the slow
// pipe), which leaves the 146 MB stream over d_Y as the bound.
//   * the weight slab W[k-slice][c0:c0+64] is split ONCE per workgroup into planes laid out as MFMA B operands in LDS: entry
//     ((step * 3 + plane) * 4 + e) * 64 + lane = the 8 weights k = 32 step + 8 (lane >> 4) .. + 7 of output column 4 (lane & 15) + e
//     (12 KB per 32-k step; a 416-row slice = 156 KB);
//   * a lane's 8 consecutive floats of d_Y (row lane & 15, k = 8 (lane >> 4) .. + 7 of the step: the 32-byte load of the fp32 version)
//     ARE its A operand of the 16x16x32 MFMA once split (txe_gemm_split.h split3x8, in registers);
//   * a wave owns TWO 16-row blocks per pass: the 12 B fragments of a step are read from LDS once for 48 MFMAs;
//   * the whole fp32 domain (txe_gemm_split.h): a weight group that holds +-Inf / NaN / |w| >= 2^120 gets the NaN marker plane, such
//     elements of d_Y poison the accumulators by themselves (Inf - Inf in the residual); a wave whose accumulators are not finite after
//     its k-slice recomputes its two row blocks with fp32 FMAs from global memory (wave-local: no LDS, no barrier).
__global__ __launch_bounds__(512) void gat_dx_pos_kernel(const DxPosArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 Bp[];    // [steps][3 planes][4 column blocks][64 lanes]
    const int ks = blockIdx.x % a.KS, rg = blockIdx.x / a.KS;
    const int k0 = ks * DXPOS_KSL, slice_len = min(DXPOS_KSL, a.K - k0);          // (a multiple of 32)
    const int nsteps = slice_len / 32;
    for (int idx = threadIdx.x; idx < nsteps * 256; idx += 512) {
        const int lane = idx & 63, e = (idx >> 6) & 3, st = idx >> 8;
        // (column vectors past the matrix re-read its last one: their products land in output columns that are never stored)
        const float* src = a.Wp + (long long)(k0 + 32 * st + 8 * (lane >> 4)) * a.ld_w + min(a.c0 + 4 * (lane & 15), a.Kp - 4) + e;
        float x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = src[(long long)t * a.ld_w];
        uint4 w1, w2, w3;
        if (split_exceptional8(x)) { split3x8(x, w1, w2, w3); w3 = make_uint4(SPL_RAW_MARK, SPL_RAW_MARK, SPL_RAW_MARK, SPL_RAW_MARK); }
        else split3x8(x, w1, w2, w3);
        uint4* d = Bp + ((st * 3) * 4 + e) * 64 + lane;
        d[0] = w1; d[256] = w2; d[512] = w3;
    }
    __syncthreads();

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int i = l & 15, kq = l >> 4;
    const int nrb = dxpos_blocks(a.n_rows);
    const int base = nrb / a.RG, rem = nrb % a.RG;
    const int rb_begin = rg * base + min(rg, rem), rb_end = rb_begin + base + (rg < rem ? 1 : 0);
    const int NG = (nsteps + 1) / 2;                               // groups of two steps (64 k): the unit of the d_Y prefetch
    float* const slab = a.part + (long long)ks * nrb * DXPOS_ROWS * DXPOS_MAXC;

    auto row_ptr = [&](int rb) {                                   // rows past the end re-read the last one (their partial rows are never used)
        const int row = min(rb * DXPOS_ROWS + i, a.n_rows - 1);
        return a.dY + (long long)row * a.ld_dy + k0 + 8 * kq;
    };
    // the second step of a slice's last group may not exist: its loads re-read the step before (never multiplied)
    auto load_group = [&](const float* ap, int g, float4 (&d)[4]) {
        const int o0 = 64 * g, o1 = min(64 * g + 32, 32 * (nsteps - 1));
        d[0] = *reinterpret_cast<const float4*>(ap + o0); d[1] = *reinterpret_cast<const float4*>(ap + o0 + 4);
        d[2] = *reinterpret_cast<const float4*>(ap + o1); d[3] = *reinterpret_cast<const float4*>(ap + o1 + 4);
    };
    int rb = rb_begin + 2 * w;
    if (rb >= rb_end) return;
    const float *apA = row_ptr(rb), *apB = row_ptr(min(rb + 1, rb_end - 1));
    float4 cA[4], cB[4], nA[4], nB[4];
    load_group(apA, 0, cA);
    load_group(apB, 0, cB);
    for (; rb < rb_end; rb += 16) {
        f32x4 acc[2][4];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[x][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int rbn = rb + 16 < rb_end ? rb + 16 : rb;            // this wave's next pair of row blocks (none left: this one again)
        const float *apAn = row_ptr(rbn), *apBn = row_ptr(min(rbn + 1, rb_end - 1));
        for (int g = 0; g < NG; ++g) {
            const bool more = g + 1 < NG;
            load_group(more ? apA : apAn, more ? g + 1 : 0, nA);
            load_group(more ? apB : apBn, more ? g + 1 : 0, nB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int st = 2 * g + h;
                if (st < nsteps) {                                 // (wave-uniform)
                    const float xa[8] = {cA[2 * h].x, cA[2 * h].y, cA[2 * h].z, cA[2 * h].w, cA[2 * h + 1].x, cA[2 * h + 1].y, cA[2 * h + 1].z, cA[2 * h + 1].w};
                    const float xb[8] = {cB[2 * h].x, cB[2 * h].y, cB[2 * h].z, cB[2 * h].w, cB[2 * h + 1].x, cB[2 * h + 1].y, cB[2 * h + 1].z, cB[2 * h + 1].w};
                    uint4 pa[3], pb[3];
                    split3x8(xa, pa[0], pa[1], pa[2]);
                    split3x8(xb, pb[0], pb[1], pb[2]);
                    const uint4* bs = Bp + (st * 12) * 64 + l;
#define TXE_DX_PROD(pl_a_, pl_b_)                                                                                          \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                       \
        const dx_bf16x8 bw = __builtin_bit_cast(dx_bf16x8, bs[((pl_b_) * 4 + e) * 64]);                                    \
        acc[0][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dx_bf16x8, pa[pl_a_]), bw, acc[0][e], 0, 0, 0); \
        acc[1][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dx_bf16x8, pb[pl_a_]), bw, acc[1][e], 0, 0, 0); \
    }
                    TXE_DX_PROD(2, 0) TXE_DX_PROD(0, 2) TXE_DX_PROD(1, 1) TXE_DX_PROD(1, 0) TXE_DX_PROD(0, 1) TXE_DX_PROD(0, 0)
#undef TXE_DX_PROD
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { cA[q] = nA[q]; cB[q] = nB[q]; }
        }
        // accumulator register r of column block e: row 4*(lane >> 4) + r, output column 4*(lane & 15) + e
        bool odd = false;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) odd |= not_finite(acc[x][e][r]);
        if (__ballot(odd) != 0ull) {                               // (rare) fp32 FMAs over the slice, straight from global memory
#pragma unroll
            for (int x = 0; x < 2; ++x) {                          // (x, r unrolled: they index accumulator registers)
                const int rbx = min(rb + x, rb_end - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* dy = a.dY + (long long)min(rbx * DXPOS_ROWS + 4 * kq + r, a.n_rows - 1) * a.ld_dy + k0;
                    const float* wc = a.Wp + (long long)k0 * a.ld_w + min(a.c0 + 4 * i, a.Kp - 4);
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
                    for (int k = 0; k < slice_len; ++k) {
                        const float av = dy[k];
                        const float4 wv = *reinterpret_cast<const float4*>(wc + (long long)k * a.ld_w);
                        sum.x = __builtin_fmaf(av, wv.x, sum.x); sum.y = __builtin_fmaf(av, wv.y, sum.y);
                        sum.z = __builtin_fmaf(av, wv.z, sum.z); sum.w = __builtin_fmaf(av, wv.w, sum.w);
                    }
                    acc[x][0][r] = sum.x; acc[x][1][r] = sum.y; acc[x][2][r] = sum.z; acc[x][3][r] = sum.w;
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (rb + x >= rb_end) break;
            float* dst = slab + ((long long)(rb + x) * DXPOS_ROWS + 4 * kq) * DXPOS_MAXC + 4 * i;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(dst + r * DXPOS_MAXC) = make_float4(acc[x][0][r], acc[x][1][r], acc[x][2][r], acc[x][3][r]);
        }
        apA = apAn; apB = apBn;
    }
}

int dxpos_prepare(DxPosArgs& a) {
    if (!a.dY || !a.Wp || !a.pos || !a.ppart || !a.part || (a.K & 127) || a.K < 128 || a.NC < 1 || a.NC > DXPOS_MAXC || (a.c0 & 3) || (a.Kp & 3) ||
        a.Kp < 4 || (a.ld_dy & 3) || (a.ld_w & 3) || (a.ld_dx & 3) || a.pcol0 < 0 || a.pcol0 + a.Pd > DXPOS_MAXC || a.vocab < 1 || a.Pd < 1)
        return TXE_ERR_ARG;
    if (!a.mask_on) { a.mask = reinterpret_cast<const unsigned*>(a.Wp); a.mask_ld = 1; a.drop_scale = 1.f; }
    a.KS = dxpos_kslices(a.K);
    int rg = device_cu_count() / a.KS;                            // one workgroup (up to 156 KB of LDS) per CU
    const int nrb = dxpos_blocks(a.n_rows);
    if (rg > nrb) rg = nrb;
    a.RG = rg < 1 ? 1 : rg;
    return TXE_OK;
}

int dxpos_launch(const DxPosArgs& a, hipStream_t stream) {
    if (a.n_rows <= 0) return TXE_OK;
    // the kernel needs more than the default 64 KB of LDS.  Function attributes are per DEVICE and this library keeps no state: set
    // before every launch (a host-side table write, idempotent, thread-safe)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gat_dx_pos_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (DXPOS_KSL / 32) * DXPOS_STEP_BYTES) != hipSuccess)
        return TXE_ERR_LAUNCH;
    // algorithmic bytes: d_Y once, the weight slab once, the outputs once
    ProfScope prof("gat_dx_pos_kernel", stream, 4.0 * ((double)a.n_rows * a.K + (double)a.K * a.NC + (double)a.n_rows * a.NC), 1);
    const int slice = a.K < DXPOS_KSL ? a.K : DXPOS_KSL;
    hipLaunchKernelGGL(gat_dx_pos_kernel, dim3(a.KS * a.RG), dim3(512), (size_t)(slice / 32) * DXPOS_STEP_BYTES, stream, a);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // namespace txe
