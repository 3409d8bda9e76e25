// GCN message/reduce (the PGCN / GCN variant), model_zoo.py:38-49 and :157-161:
//     norm = in_degree^-1/2 (inf -> 0);  out = act( norm[v] * sum_{e=(u->v)} norm[u] * hw[u]  + bias )
// Same wave-per-destination row gather as the GAT kernel (txe_gather.h) with per-source weights norm[u].
#include "txe_gather.h"
#include "txe_colsum.h"

namespace txe {

__global__ void gcn_norm_kernel(const int* __restrict__ rowptr, int n, float* __restrict__ norm) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int deg = rowptr[v + 1] - rowptr[v];
    norm[v] = deg > 0 ? 1.0f / sqrtf((float)deg) : 0.f;
}

// mode 0: out[v] = act(norm[v] * sum_u norm[u] x[u] + bias)      (forward; rowptr/col = destination CSR)
// mode 1: out[u] = norm[u] * sum_v norm[v] x[v]                   (backward; rowptr/col = source CSR)
template <int VEC, int NI>
__global__ __launch_bounds__(GAT_WAVES * 64) void gcn_aggregate_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                                       const int n_nodes, const float* __restrict__ x,
                                                                       const long long ld_x, const float* __restrict__ norm,
                                                                       const float* __restrict__ bias, const int has_act,
                                                                       const float act_slope, const int F, float* __restrict__ out,
                                                                       const long long ld_out, const int pad_to) {
    __shared__ float s_w[GAT_WAVES][64];
    __shared__ int s_idx[GAT_WAVES][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int v = xcd_remap(blockIdx.x, gridDim.x) * GAT_WAVES + w;
    if (v >= n_nodes) return;
    const int beg = rowptr[v], end = rowptr[v + 1];
    const float nv = norm[v];
    const int nvec = F / VEC;
    for (int t0 = 0; t0 < nvec; t0 += 64 * NI) {
        int hidx[NI];
        float acc[NI][VEC];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            hidx[i] = 0;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
        }
        for (int cb = beg; cb < end; cb += 64) {
            const int p = cb + l;
            if (p < end) {
                const int u = col[p];
                s_idx[w][l] = u;
                s_w[w][l] = norm[u];
            }
            __builtin_amdgcn_wave_barrier();
            gather_rows<VEC, NI, (NI >= 8 ? 1 : 2)>(x, ld_x, s_idx[w], s_w[w], min(64, end - cb), t0, nvec, hidx, acc);
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t0 + l + 64 * i;
            if (j < nvec) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float r = acc[i][k] * nv;
                    if (bias) r += bias[j * VEC + k];
                    if (has_act) r = leaky(r, act_slope);
                    acc[i][k] = r;
                }
                vstore<VEC>(out + (long long)v * ld_out + (long long)j * VEC, acc[i]);
            }
        }
    }
    for (int c = F + l; c < pad_to; c += 64) out[(long long)v * ld_out + c] = 0.f;      // (the row's padding columns, when asked for)
}

static inline int gcn_pick_vec(int F, long long ld1, long long ld2, const void* p1, const void* p2) {
    auto al = [](const void* p, int bytes) { return ((uintptr_t)p % bytes) == 0; };
    if (F % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0 && al(p1, 16) && al(p2, 16)) return 4;
    if (F % 2 == 0 && ld1 % 2 == 0 && ld2 % 2 == 0 && al(p1, 8) && al(p2, 8)) return 2;
    return 1;
}

static int gcn_launch(const int* rowptr, const int* col, int n, const float* x, long long ld_x, const float* norm,
                      const float* bias, int has_act, float slope, int F, float* out, long long ld_out, hipStream_t s, int pad_to = 0) {
    const int nb = (n + GAT_WAVES - 1) / GAT_WAVES;
    int vec = gcn_pick_vec(F, ld_x, ld_out, x, out);
    const int ni = pick_ni(F / vec);
    char kn[64];
    snprintf(kn, sizeof(kn), "gcn_aggregate_kernel<%d, %d>", vec, ni);
    ProfScope prof(kn, s, 4.0 * (2.0 * n * (double)F + 2.0 * n + 1), 1);
#define TXE_L(V, I)                                                                                                           \
    hipLaunchKernelGGL((gcn_aggregate_kernel<V, I>), dim3(nb), dim3(GAT_WAVES * 64), 0, s, rowptr, col, n, x, ld_x, norm, bias, \
                       has_act, slope, F, out, ld_out, pad_to)
    if (vec == 4) { if (ni == 8) TXE_L(4, 8); else if (ni == 4) TXE_L(4, 4); else TXE_L(4, 2); }
    else if (vec == 2) { if (ni == 8) TXE_L(2, 8); else if (ni == 4) TXE_L(2, 4); else TXE_L(2, 2); }
    else { if (ni == 8) TXE_L(1, 8); else if (ni == 4) TXE_L(1, 4); else TXE_L(1, 2); }
#undef TXE_L
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // namespace txe

using namespace txe;

extern "C" {

int txe_gcn_norm(const int* rowptr_in, int n_nodes, float* norm, void* stream) {
    if (n_nodes < 0 || !rowptr_in || !norm) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    hipLaunchKernelGGL(gcn_norm_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowptr_in, n_nodes, norm);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// out = act(norm[v] * sum_{u->v} norm[u] hw[u] + bias); has_act = 0 -> identity (last layer, model_zoo.py:152)
int txe_gcn_aggregate_fwd(const int* rowptr_in, const int* col_src, int n_nodes, const float* hw, long long ld_hw,
                          const float* norm, const float* bias, int has_act, float act_slope, int F, float* out, long long ld_out,
                          void* stream) {
    if (n_nodes < 0 || F < 1 || !rowptr_in || !hw || !norm || !out) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    return gcn_launch(rowptr_in, col_src, n_nodes, hw, ld_hw, norm, bias, has_act, act_slope, F, out, ld_out, (hipStream_t)stream);
}

size_t txe_gcn_aggregate_bwd_ws_bytes(int n_nodes, int F) { return colsum_ws_bytes(n_nodes, F); }

// d_pre: gradient w.r.t. the pre-activation output (caller applies leaky' first, e.g. txe_leaky_relu_bwd).
// d_hw[u] = norm[u] * sum_{u->v} norm[v] d_pre[v];  d_bias = column sum of d_pre (may be NULL).  The columns [F, min(ld_dhw, roundup(F, 32)))
// of d_hw are set to 0 (txe_gcn_dense_bwd wants zero padding columns).
int txe_gcn_aggregate_bwd(const int* rowptr_out, const int* col_dst, int n_nodes, const float* d_pre, long long ld_dpre,
                          const float* norm, int F, float* d_hw, long long ld_dhw, float* d_bias, void* ws, size_t ws_bytes,
                          void* stream) {
    if (n_nodes < 0 || F < 1 || !rowptr_out || !d_pre || !norm || !d_hw) return TXE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n_nodes > 0) {
        // (d_hw's padding columns up to the next multiple of 32 -- what the dense backward's GEMMs read as zeros -- are written here too)
        const long long pad = ((F + 31) / 32) * 32;
        int rc = gcn_launch(rowptr_out, col_dst, n_nodes, d_pre, ld_dpre, norm, nullptr, 0, 1.f, F, d_hw, ld_dhw, s,
                            (int)(pad < ld_dhw ? pad : ld_dhw));
        if (rc) return rc;
    }
    if (d_bias) {
        if (!ws || ws_bytes < txe_gcn_aggregate_bwd_ws_bytes(n_nodes, F)) return TXE_ERR_WORKSPACE;
        const int rc = colsum_launch(d_pre, ld_dpre, n_nodes, F, (float*)ws, d_bias, s);
        if (rc) return rc;
    }
    return TXE_OK;
}

}  // extern "C"
