// Wave-per-node weighted row gather shared by the GAT and GCN message/reduce kernels (gfx950).
#pragma once
#include "txe_common.h"

namespace txe {

constexpr int GAT_MAXH = 16;     // heads supported by the LDS staging
constexpr int GAT_WAVES = 4;     // waves (= destination nodes) per workgroup

template <int VEC> struct vec_t;
template <> struct vec_t<4> { typedef float4 type; };
template <> struct vec_t<2> { typedef float2 type; };
template <> struct vec_t<1> { typedef float type; };

template <int VEC>
__device__ __forceinline__ void vload(const float* p, float* v) {
    if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if constexpr (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    else { v[0] = *p; }
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float* v) {
    if constexpr (VEC == 4) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    else if constexpr (VEC == 2) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
    else { *p = v[0]; }
}

// acc[i] += sum_{e<cnt} w[head(i)][e] * rows[idx[e]][(j0 + lane + 64 i) * VEC ...]      for the vectors j0 <= j < j1 of the row
//
// Lane l owns vectors j0 + l + 64 i, i < NI.  Every load is UNCONDITIONAL (out-of-range lanes re-read vector j0 and their
// accumulators are never stored) and all NI*EU loads of a step are issued before the first use: hipcc puts an
// `s_waitcnt vmcnt(0)` behind every predicated load, which left ONE 16-byte load in flight per wave (measured 2.5 TB/s);
// NI*EU independent loads per lane is what fills the HBM pipe.  Used by one wave for a whole row (j0 = tile start, j1 = row
// end) and by the workgroup-cooperative kernels for a quarter row per wave.
template <int VEC, int NI, int EU>
__device__ __forceinline__ void gather_step(const float* __restrict__ base, long long ld, const int* s_idx, const float* s_w, int e,
                                            int j0, int j1, const int* hidx, float (&acc)[NI][VEC]) {
    const int l = threadIdx.x & 63;
    float v[EU][NI][VEC];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
        const float* row = base + (long long)s_idx[e + u] * ld;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = j0 + l + 64 * i;
            const int jc = (j < j1) ? j : j0;                   // clamped: loads stay unconditional
            vload<VEC>(row + (long long)jc * VEC, v[u][i]);
        }
    }
#pragma unroll
    for (int u = 0; u < EU; ++u)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float a = s_w[hidx[i] * 64 + e + u];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[i][k] = fmaf(a, v[u][i][k], acc[i][k]);
        }
}

template <int VEC, int NI, int EU>
__device__ __forceinline__ void gather_rows(const float* __restrict__ base, long long ld, const int* s_idx,
                                            const float* s_w, int cnt, int j0, int j1, const int* hidx,
                                            float (&acc)[NI][VEC]) {
    int e = 0;
    for (; e + EU <= cnt; e += EU) gather_step<VEC, NI, EU>(base, ld, s_idx, s_w, e, j0, j1, hidx, acc);
    if constexpr (EU >= 4) {                                    // remainder in halves: 2 edges still go out together
        if (e + 2 <= cnt) { gather_step<VEC, NI, 2>(base, ld, s_idx, s_w, e, j0, j1, hidx, acc); e += 2; }
    }
    for (; e < cnt; ++e) gather_step<VEC, NI, 1>(base, ld, s_idx, s_w, e, j0, j1, hidx, acc);
}

// vectors-per-lane template choice for a row of nvec vectors: 2 / 4 / 8 (wider rows loop over 512-vector tiles)
static inline int pick_ni(int nvec) { return nvec <= 128 ? 2 : (nvec <= 256 ? 4 : 8); }

}  // namespace txe
