// Graph readouts of the batched egonets (model_zoo.py:227-242):
//   MeanReadout          hg[g] = mean_v h[v]                                  (dgl.mean_nodes, :232)
//   WeightedMeanReadout  w_v = softplus(position_weights[pos_v]);             (:241)
//                        hg[g] = sum_v w_v h[v] / sum_v w_v                   (:242)
// One wavefront per egonet: lanes span the feature row with 16-byte loads, the (few) nodes of the egonet
// are walked serially; HBM-bound, every h row is read exactly once.  Backward is atomic free: the three
// position-weight gradients are reduced per egonet, then over egonets in a second tiny kernel.
#include "txe_common.h"

namespace txe {

constexpr int RO_WAVES = 4;
constexpr int RO_MAX_VOCAB = 8;

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(__expf(x)); }   // F.softplus, threshold 20
__device__ __forceinline__ float sigmoid_t(float x) { return x > 20.f ? 1.f : 1.f / (1.f + __expf(-x)); }

template <int VEC>
__device__ __forceinline__ void ro_vload(const float* p, float* v) {
    if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if constexpr (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    else { v[0] = *p; }
}
template <int VEC>
__device__ __forceinline__ void ro_vstore(float* p, const float* v) {
    if constexpr (VEC == 4) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    else if constexpr (VEC == 2) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
    else { *p = v[0]; }
}

// per-node weights of a <=64-node chunk, one node per lane: w = softplus(pw[pos]) (or 1), parked in LDS for the row sweep
__device__ __forceinline__ float node_weight(const int* __restrict__ pos, const float* __restrict__ pw, int v, bool valid) {
    if (!valid) return 0.f;
    return pw ? softplus_t(pw[pos[v]]) : 1.f;
}

// Lane l owns vectors t0 + l + 64 i (i < NI) of the row; EU nodes are swept together so that NI*EU independent 16-byte loads
// are in flight per lane (all loads unconditional: out-of-range vectors / nodes are clamped and weighted 0).
template <int NI> struct ro_eu { static constexpr int value = NI >= 8 ? 1 : (NI == 4 ? 2 : 4); };

template <int VEC, int NI>
__global__ __launch_bounds__(RO_WAVES * 64) void readout_fwd_kernel(const int* __restrict__ goff, const int G,
                                                                    const float* __restrict__ h, const long long ld_h,
                                                                    const int* __restrict__ pos, const float* __restrict__ pw,
                                                                    const int D, float* __restrict__ hg, float* __restrict__ wsum) {
    constexpr int EU = ro_eu<NI>::value;
    __shared__ float s_w[RO_WAVES][64 + EU];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g = xcd_remap(blockIdx.x, gridDim.x) * RO_WAVES + w;
    if (g >= G) return;
    const int beg = goff[g], end = goff[g + 1];
    const int nvec = D / VEC;
    if (l < EU) s_w[w][64 + l] = 0.f;                       // the unrolled sweep may look EU - 1 nodes past a full chunk
    for (int t0 = 0; t0 < nvec; t0 += 64 * NI) {
        int off[NI];
        float acc[NI][VEC];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t0 + l + 64 * i;
            off[i] = ((j < nvec) ? j : t0) * VEC;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
        }
        float S = 0.f;
        for (int cb = beg; cb < end; cb += 64) {
            const float wl = node_weight(pos, pw, cb + l, cb + l < end);
            S += wl;
            s_w[w][l] = wl;
            __builtin_amdgcn_wave_barrier();
            const int cnt = min(64, end - cb);
            for (int e = 0; e < cnt; e += EU) {
                float x[EU][NI][VEC];
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const float* row = h + (long long)min(cb + e + u, end - 1) * ld_h;
#pragma unroll
                    for (int i = 0; i < NI; ++i) ro_vload<VEC>(row + off[i], x[u][i]);
                }
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const float wv = s_w[w][e + u];             // 0 for nodes past the end of the egonet
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[i][k] = fmaf(wv, x[u][i][k], acc[i][k]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        S = wave_sum(S);
        if (l == 0 && wsum && t0 == 0) wsum[g] = S;
        const float inv = 1.f / S;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t0 + l + 64 * i;
            if (j < nvec) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[i][k] *= inv;
                ro_vstore<VEC>(hg + (long long)g * D + (long long)j * VEC, acc[i]);
            }
        }
    }
}

// d_h[v] = (w_v / S_g) d_hg[g];   d_w_v = <d_hg[g], h[v] - hg[g]> / S_g;   d_pw[c] += d_w_v * sigmoid(pw[c]) for pos_v == c
template <int VEC, int NI>
__global__ __launch_bounds__(RO_WAVES * 64) void readout_bwd_kernel(const int* __restrict__ goff, const int G,
                                                                    const float* __restrict__ h, const long long ld_h,
                                                                    const int* __restrict__ pos, const float* __restrict__ pw,
                                                                    const int vocab, const int D, const float* __restrict__ hg,
                                                                    const float* __restrict__ wsum, const float* __restrict__ d_hg,
                                                                    float* __restrict__ d_h, const long long ld_dh,
                                                                    float* __restrict__ dpw_part /*[G][vocab]*/) {
    constexpr int EU = ro_eu<NI>::value;
    __shared__ float s_sc[RO_WAVES][64 + EU];      // w_v / S
    __shared__ float s_sg[RO_WAVES][64 + EU];      // sigmoid(pw[pos_v]) / S
    __shared__ int s_pc[RO_WAVES][64 + EU];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g = xcd_remap(blockIdx.x, gridDim.x) * RO_WAVES + w;
    if (g >= G) return;
    const int beg = goff[g], end = goff[g + 1];
    const float inv = 1.f / wsum[g];
    const int nvec = D / VEC;
    if (l < EU) { s_sc[w][64 + l] = 0.f; s_sg[w][64 + l] = 0.f; s_pc[w][64 + l] = 0; }
    float dpw[RO_MAX_VOCAB];
#pragma unroll
    for (int c = 0; c < RO_MAX_VOCAB; ++c) dpw[c] = 0.f;
    const float* dgrow = d_hg + (long long)g * D;
    const float* mrow = hg + (long long)g * D;
    for (int cb = beg; cb < end; cb += 64) {
        {
            const int v = cb + l;
            const bool valid = v < end;
            const int pc = (valid && pw) ? pos[v] : 0;
            const float x = pw ? pw[pc] : 0.f;
            s_pc[w][l] = pc;
            s_sc[w][l] = valid ? (pw ? softplus_t(x) : 1.f) * inv : 0.f;
            s_sg[w][l] = (valid && pw) ? sigmoid_t(x) * inv : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        const int cnt = min(64, end - cb);
        for (int e = 0; e < cnt; e += EU) {
            float part[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) part[u] = 0.f;
            for (int t0 = 0; t0 < nvec; t0 += 64 * NI) {
                int off[NI];
                float dg[NI][VEC], m[NI][VEC], x[EU][NI][VEC];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int j = t0 + l + 64 * i;
                    off[i] = ((j < nvec) ? j : t0) * VEC;
                    ro_vload<VEC>(dgrow + off[i], dg[i]);
                    if (pw) ro_vload<VEC>(mrow + off[i], m[i]);
                }
                if (pw) {
#pragma unroll
                    for (int u = 0; u < EU; ++u) {
                        const float* row = h + (long long)min(cb + e + u, end - 1) * ld_h;
#pragma unroll
                        for (int i = 0; i < NI; ++i) ro_vload<VEC>(row + off[i], x[u][i]);
                    }
                }
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int v = cb + e + u;
                    const float sc = s_sc[w][e + u];
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const bool live = (t0 + l + 64 * i) < nvec;
                        if (live && v < end) {
                            float o[VEC];
#pragma unroll
                            for (int k = 0; k < VEC; ++k) o[k] = sc * dg[i][k];
                            ro_vstore<VEC>(d_h + (long long)v * ld_dh + off[i], o);
                        }
                        if (pw) {
                            float d = 0.f;
#pragma unroll
                            for (int k = 0; k < VEC; ++k) d = fmaf(dg[i][k], x[u][i][k] - m[i][k], d);
                            part[u] += live ? d : 0.f;
                        }
                    }
                }
            }
            if (pw) {
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const float pt = wave_sum(part[u]) * s_sg[w][e + u];      // 0 past the end of the egonet
                    const int pc = s_pc[w][e + u];
#pragma unroll
                    for (int c = 0; c < RO_MAX_VOCAB; ++c) dpw[c] += (pc == c) ? pt : 0.f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (pw && l == 0) {
#pragma unroll
        for (int c = 0; c < RO_MAX_VOCAB; ++c)
            if (c < vocab) dpw_part[(long long)g * vocab + c] = dpw[c];
    }
}

__global__ void readout_dpw_reduce_kernel(const float* __restrict__ part, int G, int vocab, float* __restrict__ d_pw) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    for (int g = threadIdx.x; g < G; g += blockDim.x) acc += part[(long long)g * vocab + c];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) d_pw[c] = red[0] + red[1] + red[2] + red[3];
}

// ---- SumReadout / MaxReadout / ConcatReadout (model_zoo.py:244-276) ---------------------------------------------------
// mode 1 SUM   : hg[g][d]       = sum_v h[v][d]
// mode 2 MAX   : hg[g][d]       = max_v h[v][d]            (argmax[g][d] = first maximiser, for backward)
// mode 3 CONCAT: hg[g][c*D + d] = sum_{v: pos_v == c} h[v][d] * s_c,  s_0 = s_2 = 1/n_g,  s_1 = 1/#{pos == 1}   (c < 3)
// One wavefront per egonet; a lane owns RM_NB columns 64 apart, so every node costs RM_NB independent loads (first version: one
// column at a time, one load in flight -- 126 us for the MAG batch; these variants are API completeness, not the hot path).
constexpr int RM_NB = 8;
template <int mode>
__global__ __launch_bounds__(RO_WAVES * 64) void readout_multi_fwd_kernel(const int* __restrict__ goff, const int G,
                                                                          const float* __restrict__ h, const long long ld_h,
                                                                          const int* __restrict__ pos, const int D,
                                                                          float* __restrict__ hg, int* __restrict__ argmax) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g = blockIdx.x * RO_WAVES + w;
    if (g >= G) return;
    const int beg = goff[g], end = goff[g + 1];
    float cnt1 = 0.f;
    if (mode == 3) {
        for (int v = beg + l; v < end; v += 64) cnt1 += (pos[v] == 1) ? 1.f : 0.f;
        cnt1 = wave_sum(cnt1);
    }
    const float inv_n = 1.f / (float)(end - beg), inv_1 = 1.f / cnt1;
    for (int d0 = 0; d0 < D; d0 += 64 * RM_NB) {
        int dc[RM_NB];
        float a0[RM_NB], a1[RM_NB], a2[RM_NB];
        int am[RM_NB];
#pragma unroll
        for (int i = 0; i < RM_NB; ++i) {
            const int d = d0 + l + 64 * i;
            dc[i] = (d < D) ? d : 0;                            // clamped: loads stay unconditional, results of dead slots are dropped
            a0[i] = (mode == 2) ? -INFINITY : 0.f;
            a1[i] = 0.f; a2[i] = 0.f; am[i] = beg;
        }
        for (int v = beg; v < end; ++v) {
            const int pc = (mode == 3) ? pos[v] : 0;
            float x[RM_NB];
#pragma unroll
            for (int i = 0; i < RM_NB; ++i) x[i] = h[(long long)v * ld_h + dc[i]];
#pragma unroll
            for (int i = 0; i < RM_NB; ++i) {
                if (mode == 1) a0[i] += x[i];
                else if (mode == 2) { if (x[i] > a0[i]) { a0[i] = x[i]; am[i] = v; } }
                else { a0[i] += (pc == 0) ? x[i] : 0.f; a1[i] += (pc == 1) ? x[i] : 0.f; a2[i] += (pc == 2) ? x[i] : 0.f; }
            }
        }
#pragma unroll
        for (int i = 0; i < RM_NB; ++i) {
            const int d = d0 + l + 64 * i;
            if (d >= D) continue;
            if (mode == 3) {
                hg[(long long)g * 3 * D + d] = a0[i] * inv_n;
                hg[(long long)g * 3 * D + D + d] = a1[i] * inv_1;
                hg[(long long)g * 3 * D + 2 * D + d] = a2[i] * inv_n;
            } else {
                hg[(long long)g * D + d] = a0[i];
                if (mode == 2 && argmax) argmax[(long long)g * D + d] = am[i];
            }
        }
    }
}

template <int mode>
__global__ __launch_bounds__(RO_WAVES * 64) void readout_multi_bwd_kernel(const int* __restrict__ goff, const int G,
                                                                          const int* __restrict__ pos, const int D,
                                                                          const float* __restrict__ d_hg, const int* __restrict__ argmax,
                                                                          float* __restrict__ d_h, const long long ld_dh) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g = blockIdx.x * RO_WAVES + w;
    if (g >= G) return;
    const int beg = goff[g], end = goff[g + 1];
    float cnt1 = 0.f;
    if (mode == 3) {
        for (int v = beg + l; v < end; v += 64) cnt1 += (pos[v] == 1) ? 1.f : 0.f;
        cnt1 = wave_sum(cnt1);
    }
    const float inv_n = 1.f / (float)(end - beg), inv_1 = 1.f / cnt1;
    for (int d0 = 0; d0 < D; d0 += 64 * RM_NB) {
        // the graph's gradient row(s) are fetched once per column slot, then every node's row is a pure store
        float r0[RM_NB], r1[RM_NB], r2[RM_NB];
        int am[RM_NB];
#pragma unroll
        for (int i = 0; i < RM_NB; ++i) {
            const int d = d0 + l + 64 * i;
            const int dcl = (d < D) ? d : 0;
            if (mode == 3) {
                r0[i] = d_hg[(long long)g * 3 * D + dcl] * inv_n;
                r1[i] = d_hg[(long long)g * 3 * D + D + dcl] * inv_1;
                r2[i] = d_hg[(long long)g * 3 * D + 2 * D + dcl] * inv_n;
                am[i] = 0;
            } else {
                r0[i] = d_hg[(long long)g * D + dcl];
                r1[i] = 0.f; r2[i] = 0.f;
                am[i] = (mode == 2) ? argmax[(long long)g * D + dcl] : 0;
            }
        }
        for (int v = beg; v < end; ++v) {
            const int pc = (mode == 3) ? pos[v] : 0;
#pragma unroll
            for (int i = 0; i < RM_NB; ++i) {
                const int d = d0 + l + 64 * i;
                if (d >= D) continue;
                float r;
                if (mode == 1) r = r0[i];
                else if (mode == 2) r = (am[i] == v) ? r0[i] : 0.f;
                else r = (pc == 0) ? r0[i] : ((pc == 1) ? r1[i] : ((pc == 2) ? r2[i] : 0.f));
                d_h[(long long)v * ld_dh + d] = r;
            }
        }
    }
}

static inline int ro_pick_vec(int D, long long ld1, long long ld2, const void* a, const void* b, const void* c) {
    auto al = [](const void* p, int bytes) { return p == nullptr || ((uintptr_t)p % bytes) == 0; };
    if (D % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0 && al(a, 16) && al(b, 16) && al(c, 16)) return 4;
    if (D % 2 == 0 && ld1 % 2 == 0 && ld2 % 2 == 0 && al(a, 8) && al(b, 8) && al(c, 8)) return 2;
    return 1;
}

}  // namespace txe

using namespace txe;

extern "C" {

// pw == NULL -> MeanReadout; else WeightedMeanReadout with softplus(pw[pos]).  wsum[G] (may be NULL in
// inference) receives sum_v w_v, needed by backward.
int txe_readout_fwd(const int* graph_off, int G, const float* h, long long ld_h, const int* pos, const float* pw, int D,
                    float* hg, float* wsum, void* stream) {
    if (G < 0 || D < 1 || !graph_off || !h || !hg || (pw && !pos)) return TXE_ERR_ARG;
    if (G == 0) return TXE_OK;
    const int nb = (G + RO_WAVES - 1) / RO_WAVES;
    const int vec = ro_pick_vec(D, ld_h, D, h, hg, nullptr);
    hipStream_t s = (hipStream_t)stream;
    const int ni = D / vec <= 128 ? 2 : (D / vec <= 256 ? 4 : 8);
    char kn[64];
    snprintf(kn, sizeof(kn), "readout_fwd_kernel<%d, %d>", vec, ni);
    ProfScope prof(kn, s, 4.0 * (double)G * D, 1);   // output bytes; the caller adds the N*D input rows
#define TXE_L(V, I) hipLaunchKernelGGL((readout_fwd_kernel<V, I>), dim3(nb), dim3(RO_WAVES * 64), 0, s, graph_off, G, h, ld_h, pos, pw, D, hg, wsum)
    if (vec == 4) { if (ni == 8) TXE_L(4, 8); else if (ni == 4) TXE_L(4, 4); else TXE_L(4, 2); }
    else if (vec == 2) { if (ni == 8) TXE_L(2, 8); else if (ni == 4) TXE_L(2, 4); else TXE_L(2, 2); }
    else { if (ni == 8) TXE_L(1, 8); else if (ni == 4) TXE_L(1, 4); else TXE_L(1, 2); }
#undef TXE_L
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// dpw_ws: G*vocab floats of scratch (only when pw != NULL).
int txe_readout_bwd(const int* graph_off, int G, const float* h, long long ld_h, const int* pos, const float* pw, int vocab,
                    int D, const float* hg, const float* wsum, const float* d_hg, float* d_h, long long ld_dh, float* d_pw,
                    float* dpw_ws, void* stream) {
    if (G < 0 || D < 1 || !graph_off || !h || !hg || !wsum || !d_hg || !d_h) return TXE_ERR_ARG;
    if (pw && (!pos || !d_pw || !dpw_ws || vocab < 1 || vocab > RO_MAX_VOCAB)) return TXE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (G > 0) {
        const int nb = (G + RO_WAVES - 1) / RO_WAVES;
        int vec = ro_pick_vec(D, ld_h, ld_dh, h, d_h, hg);
        if (vec > 1 && ((uintptr_t)d_hg % (vec * 4)) != 0) vec = 1;
        const int ni = D / vec <= 128 ? 2 : (D / vec <= 256 ? 4 : 8);
        char kn[64];
        snprintf(kn, sizeof(kn), "readout_bwd_kernel<%d, %d>", vec, ni);
        ProfScope prof(kn, s, 4.0 * (double)G * D, 1);   // d_hg rows; the caller adds the N*D rows read (h) and written (d_h)
#define TXE_L(V, I)                                                                                                          \
    hipLaunchKernelGGL((readout_bwd_kernel<V, I>), dim3(nb), dim3(RO_WAVES * 64), 0, s, graph_off, G, h, ld_h, pos, pw, vocab, \
                       D, hg, wsum, d_hg, d_h, ld_dh, dpw_ws)
        if (vec == 4) { if (ni == 8) TXE_L(4, 8); else if (ni == 4) TXE_L(4, 4); else TXE_L(4, 2); }
        else if (vec == 2) { if (ni == 8) TXE_L(2, 8); else if (ni == 4) TXE_L(2, 4); else TXE_L(2, 2); }
        else { if (ni == 8) TXE_L(1, 8); else if (ni == 4) TXE_L(1, 4); else TXE_L(1, 2); }
#undef TXE_L
        TXE_CHECK_LAUNCH();
    }
    if (pw) {
        hipLaunchKernelGGL(readout_dpw_reduce_kernel, dim3(vocab), dim3(256), 0, s, (const float*)dpw_ws, G, vocab, d_pw);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

// SumReadout (mode 1), MaxReadout (mode 2), ConcatReadout (mode 3) -- model_zoo.py:244-276.  hg is [G][D] ([G][3D] for mode 3);
// argmax [G][D] int32 is written in mode 2 (may be NULL in inference) and read by the backward.
int txe_readout_multi_fwd(const int* graph_off, int G, const float* h, long long ld_h, const int* pos, int D, int mode, float* hg,
                          int* argmax, void* stream) {
    if (G < 0 || D < 1 || mode < 1 || mode > 3 || !graph_off || !h || !hg || (mode == 3 && !pos)) return TXE_ERR_ARG;
    if (G == 0) return TXE_OK;
    const dim3 grid((G + RO_WAVES - 1) / RO_WAVES), block(RO_WAVES * 64);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 1) hipLaunchKernelGGL(readout_multi_fwd_kernel<1>, grid, block, 0, s, graph_off, G, h, ld_h, pos, D, hg, argmax);
    else if (mode == 2) hipLaunchKernelGGL(readout_multi_fwd_kernel<2>, grid, block, 0, s, graph_off, G, h, ld_h, pos, D, hg, argmax);
    else hipLaunchKernelGGL(readout_multi_fwd_kernel<3>, grid, block, 0, s, graph_off, G, h, ld_h, pos, D, hg, argmax);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_readout_multi_bwd(const int* graph_off, int G, const int* pos, int D, int mode, const float* d_hg, const int* argmax,
                          float* d_h, long long ld_dh, void* stream) {
    if (G < 0 || D < 1 || mode < 1 || mode > 3 || !graph_off || !d_hg || !d_h || (mode == 3 && !pos) || (mode == 2 && !argmax))
        return TXE_ERR_ARG;
    if (G == 0) return TXE_OK;
    const dim3 grid((G + RO_WAVES - 1) / RO_WAVES), block(RO_WAVES * 64);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 1) hipLaunchKernelGGL(readout_multi_bwd_kernel<1>, grid, block, 0, s, graph_off, G, pos, D, d_hg, argmax, d_h, ld_dh);
    else if (mode == 2) hipLaunchKernelGGL(readout_multi_bwd_kernel<2>, grid, block, 0, s, graph_off, G, pos, D, d_hg, argmax, d_h, ld_dh);
    else hipLaunchKernelGGL(readout_multi_bwd_kernel<3>, grid, block, 0, s, graph_off, G, pos, D, d_hg, argmax, d_h, ld_dh);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"
