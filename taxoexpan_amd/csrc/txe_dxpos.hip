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

#include <stdlib.h>

namespace txe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

int device_cu_count();     // txe_profile.hip

__global__ __launch_bounds__(512) void gat_dx_pos_kernel(const DxPosArgs a) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];    // [slice_len][64]
    const int ks = blockIdx.x % a.KS, rg = blockIdx.x / a.KS;
    const int k0 = ks * DXPOS_KSL, slice_len = min(DXPOS_KSL, a.K - k0);          // (a multiple of 128)
    {   // the weight slice: every load first, then the LDS stores (column vectors past the matrix re-read its last one: their
        // products land in output columns that are never stored)
        const int i = threadIdx.x & 15, r = threadIdx.x >> 4;
        const float* src = a.Wp + (long long)(k0 + r) * a.ld_w + min(a.c0 + 4 * i, a.Kp - 4);
        for (int n0 = 0; n0 < slice_len / 32; n0 += 4) {           // (slice_len is a multiple of 128: whole batches, no branch)
            float4 t[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) t[n] = *reinterpret_cast<const float4*>(src + (long long)(n0 + n) * 32 * a.ld_w);
#pragma unroll
            for (int n = 0; n < 4; ++n) *reinterpret_cast<float4*>(Bs + ((n0 + n) * 32 + r) * 64 + 4 * i) = t[n];
        }
    }
    __syncthreads();

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int i = l & 15, kq = l >> 4;
    const int nrb = dxpos_blocks(a.n_rows);
    const int base = nrb / a.RG, rem = nrb % a.RG;
    const int rb_begin = rg * base + min(rg, rem), rb_end = rb_begin + base + (rg < rem ? 1 : 0);
    const int NG = slice_len / 128;
    const float* bl = Bs + (8 * kq) * 64 + 4 * i;                  // this lane's part of the B addresses
    float* const slab = a.part + (long long)ks * nrb * DXPOS_ROWS * DXPOS_MAXC;

    auto row_ptr = [&](int rb) {                                   // rows past the end re-read the last one (their partial rows are never used)
        const int row = min(rb * DXPOS_ROWS + i, a.n_rows - 1);
        return a.dY + (long long)row * a.ld_dy + k0 + 8 * kq;
    };
    int rb = rb_begin + w;
    if (rb >= rb_end) return;
    const float* ap = row_ptr(rb);
    float4 ac[8], an[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ac[2 * p] = *reinterpret_cast<const float4*>(ap + 32 * p);
        ac[2 * p + 1] = *reinterpret_cast<const float4*>(ap + 32 * p + 4);
    }
    for (; rb < rb_end; rb += 8) {
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < NG; ++g) {
            // the next group: further along this row block, else the head of this wave's next row block (none left: this one again)
            const bool more = g + 1 < NG;
            const int rbn = rb + 8 < rb_end ? rb + 8 : rb;
            const float* apn = more ? ap + 128 : row_ptr(rbn);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                an[2 * p] = *reinterpret_cast<const float4*>(apn + 32 * p);
                an[2 * p + 1] = *reinterpret_cast<const float4*>(apn + 32 * p + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* bg = bl + g * 128 * 64;
            // B fragments two steps ahead of the MFMAs that use them (ds_read latency under the previous steps' MFMAs)
#define TXE_DX_ROW(j_) (32 * ((j_) >> 3) + ((j_) & 7))              /* step j = 8 p + 4 h + t  ->  weight row 32 p + 4 h + t (+ 8 kq) */
            float4 bq[3];
            bq[0] = *reinterpret_cast<const float4*>(bg + TXE_DX_ROW(0) * 64);
            bq[1] = *reinterpret_cast<const float4*>(bg + TXE_DX_ROW(1) * 64);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j + 2 < 32) bq[(j + 2) % 3] = *reinterpret_cast<const float4*>(bg + TXE_DX_ROW(j + 2) * 64);
                __builtin_amdgcn_sched_barrier(0);                 // (left alone the scheduler sinks every read to just before its use)
                const float4 v = ac[j >> 2];
                const float av = (j & 3) == 0 ? v.x : ((j & 3) == 1 ? v.y : ((j & 3) == 2 ? v.z : v.w));
                const float4 b = bq[j % 3];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.w, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef TXE_DX_ROW
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) ac[q] = an[q];
            ap = apn;
        }
        // accumulator register r of column block e: row 4*(lane >> 4) + r, output column 4*(lane & 15) + e
        float* dst = slab + ((long long)rb * DXPOS_ROWS + 4 * kq) * DXPOS_MAXC + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(dst + r * DXPOS_MAXC) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
}

int dxpos_prepare(DxPosArgs& a) {
    if (!a.dY || !a.Wp || !a.pos || !a.ppart || !a.part || (a.K & 127) || a.K < 128 || a.NC < 1 || a.NC > DXPOS_MAXC || (a.c0 & 3) || (a.Kp & 3) ||
        a.Kp < 4 || (a.ld_dy & 3) || (a.ld_w & 3) || (a.ld_dx & 3) || a.pcol0 < 0 || a.pcol0 + a.Pd > DXPOS_MAXC || a.vocab < 1 || a.Pd < 1)
        return TXE_ERR_ARG;
    if (!a.mask_on) { a.mask = reinterpret_cast<const unsigned*>(a.Wp); a.mask_ld = 1; a.drop_scale = 1.f; }
    a.KS = dxpos_kslices(a.K);
    int rg = device_cu_count() / a.KS;                            // one workgroup (128 KB of LDS) per CU
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
                            DXPOS_KSL * DXPOS_MAXC * (int)sizeof(float)) != hipSuccess)
        return TXE_ERR_LAUNCH;
    // algorithmic bytes: d_Y once, the weight slab once, the outputs once
    ProfScope prof("gat_dx_pos_kernel", stream, 4.0 * ((double)a.n_rows * a.K + (double)a.K * a.NC + (double)a.n_rows * a.NC), 1);
    const int slice = a.K < DXPOS_KSL ? a.K : DXPOS_KSL;
    hipLaunchKernelGGL(gat_dx_pos_kernel, dim3(a.KS * a.RG), dim3(512), (size_t)slice * DXPOS_MAXC * sizeof(float), stream, a);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // namespace txe
