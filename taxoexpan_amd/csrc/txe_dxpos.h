// First-layer d_X (position columns only) as ONE HBM stream over d_Y -- see txe_dxpos.hip.
#pragma once
#include "txe_common.h"

namespace txe {

constexpr int DXPOS_ROWS = 16;       // rows of d_Y per row block (= one 16x16x4 MFMA row block)
constexpr int DXPOS_MAXC = 64;       // widest column range [c0, c0 + NC) the kernel covers
constexpr int DXPOS_KSL = 416;       // k-slice of the weight slab a workgroup keeps in LDS as bf16 planes: 13 steps of 32 k
constexpr int DXPOS_STEP_BYTES = 3 * 4 * 64 * 16;   // 3 planes x 4 column blocks x 64 lanes x 16 bytes = 12 KB per step (156 KB per slice)

struct DxPosArgs {
    const float* dY; long long ld_dy; int n_rows; int K;     // d_Y [n_rows][K], K a multiple of 128
    const float* Wp; long long ld_w; int Kp;                 // Wp [K][Kp]; columns [c0, c0 + NC) are used (c0 % 4 == 0, NC <= 64)
    int c0, NC;
    const unsigned* mask; long long mask_ld; int mask_on; float drop_scale;   // keep bits of the [n_rows][Kt] layer input (always readable)
    float* dX; long long ld_dx;                              // d_X [n_rows][Kp] (columns [c0, c0 + roundup(NC, 4)) written) or NULL
    const int* pos; int vocab, Pd, pcol0;                    // position classes; ppart[b][v][j] = sum_{rows of block b, pos == v} out[row][pcol0 + j]
    float* ppart;
    float* part;                                             // [KS][nrb * 16][64] raw partial products, one slab per k-slice
    int KS, RG;                                              // k-slices, row groups (grid = KS * RG workgroups)
};

__host__ __device__ static inline int dxpos_blocks(int n_rows) { return (n_rows + DXPOS_ROWS - 1) / DXPOS_ROWS; }
static inline int dxpos_kslices(int K) { return (K + DXPOS_KSL - 1) / DXPOS_KSL; }
static inline size_t dxpos_part_bytes(int n_rows, int K) {
    return (size_t)dxpos_kslices(K) * dxpos_blocks(n_rows) * DXPOS_ROWS * DXPOS_MAXC * sizeof(float);
}
// fills a.KS / a.RG / the mask defaults; returns TXE_ERR_ARG on a shape the kernel does not cover
int dxpos_prepare(DxPosArgs& a);
// the streaming launch: a.part <- the KS partial products (everything else happens in dxpos_finish_job)
int dxpos_launch(const DxPosArgs& a, hipStream_t stream);

#if defined(__HIPCC__)
// One 256-thread workgroup per 16-row block: the KS partial products in slice order, dropout keep bits, the d_X store, and the block's
// per-class partial sums of the position columns (dP's first stage).  Runs as a job of the layer's reduction launch.
__device__ __forceinline__ void dxpos_finish_job(const int rb, const DxPosArgs& a) {
    __shared__ __attribute__((aligned(16))) float tile[DXPOS_ROWS][DXPOS_MAXC + 4];
    __shared__ int s_pos[DXPOS_ROWS];
    const int r0 = rb * DXPOS_ROWS;
    if (threadIdx.x < DXPOS_ROWS) s_pos[threadIdx.x] = (r0 + (int)threadIdx.x < a.n_rows) ? a.pos[r0 + threadIdx.x] : -1;
    const int rr = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int m = r0 + rr, mc = min(m, a.n_rows - 1);
    const int gc = a.c0 + 4 * cq;                                  // (4 consecutive columns from a multiple of 4 share a mask word)
    const unsigned mwd = a.mask[a.mask_on ? ((long long)mc * a.mask_ld + min((long long)(gc >> 5), a.mask_ld - 1)) : 0];
    const long long slab = (long long)dxpos_blocks(a.n_rows) * DXPOS_ROWS * DXPOS_MAXC;
    const float* pp = a.part + ((long long)m * DXPOS_MAXC + 4 * cq);
    float4 u[4];                                                   // (KS <= 4 in one round trip; more slices loop)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < a.KS; s0 += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = *reinterpret_cast<const float4*>(pp + (long long)min(s0 + q, a.KS - 1) * slab);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = s0 + q < a.KS;
            v.x += ok ? u[q].x : 0.f; v.y += ok ? u[q].y : 0.f; v.z += ok ? u[q].z : 0.f; v.w += ok ? u[q].w : 0.f;
        }
    }
    const unsigned kb = a.mask_on ? (mwd >> (gc & 31)) : 0xFu;
    v.x = (kb & 1u) ? v.x * a.drop_scale : 0.f;
    v.y = (kb & 2u) ? v.y * a.drop_scale : 0.f;
    v.z = (kb & 4u) ? v.z * a.drop_scale : 0.f;
    v.w = (kb & 8u) ? v.w * a.drop_scale : 0.f;
    if (a.dX != nullptr && m < a.n_rows && 4 * cq < a.NC) *reinterpret_cast<float4*>(a.dX + (long long)m * a.ld_dx + gc) = v;
    *reinterpret_cast<float4*>(&tile[rr][4 * cq]) = v;
    __syncthreads();
    for (int t = threadIdx.x; t < a.vocab * a.Pd; t += 256) {
        const int cls = t / a.Pd, j = t - cls * a.Pd;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < DXPOS_ROWS; ++r) s += (s_pos[r] == cls) ? tile[r][a.pcol0 + j] : 0.f;
        a.ppart[((long long)rb * a.vocab + cls) * a.Pd + j] = s;
    }
}
#endif

}  // namespace txe
