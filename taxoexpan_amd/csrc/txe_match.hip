// Bilinear matchers and the all-candidate scoring loop.
//   BIM / LBM  (model_zoo.py:301-328):  s_i = e1_i^T W e2_i   (nn.Bilinear(l, r, 1, bias=False)),  LBM: exp(s_i)
//   scoring loop (test_fast.py:116-123, infer.py:95-99): for every query q, match(hg, q.expand(G,-1)) -> G scores.
// The literal loop costs 2*l*r flops per (query, candidate) pair.  Here the bilinear form is factored once,
// U = HG W  (G x r), and every query block is one NT GEMM  S[q][g] = <Q[q], U[g]>  with the exp fused into the
// epilogue -- 2*r flops per pair on the fp32 MFMA pipe.
#include "txe_gemm.h"

namespace txe {

// s[i] = <U[i], e2[i]>  (exp optionally); one wavefront per row.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ U, const float* __restrict__ e2, long long ld_e2,
                                                     int G, int r, int apply_exp, float* __restrict__ s) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + w;
    if (i >= G) return;
    float acc = 0.f;
    for (int k = l; k < r; k += 64) acc = fmaf(U[(long long)i * r + k], e2[(long long)i * ld_e2 + k], acc);
    acc = wave_sum(acc);
    if (l == 0) s[i] = apply_exp ? __expf(acc) : acc;
}

// dsl[i] = ds[i] * (apply_exp ? s[i] : 1)
__global__ void dsl_kernel(const float* __restrict__ ds, const float* __restrict__ s, int apply_exp, int G, float* __restrict__ dsl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G) dsl[i] = apply_exp ? ds[i] * s[i] : ds[i];
}

// dU[i][k] = dsl[i] * e2[i][k]      (gradient of U = E1 W, the only consumer of e2 in s_i = <U_i, e2_i>)
__global__ void du_kernel(const float* __restrict__ dsl, const float* __restrict__ e2, long long ld_e2, int G, int r,
                          float* __restrict__ dU) {
    const long long n = (long long)G * r;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / r;
        const int k = (int)(t % r);
        dU[t] = dsl[i] * e2[i * ld_e2 + k];
    }
}

// d_e2[i][k] = dsl[i] * U[i][k]
__global__ void de2_kernel(const float* __restrict__ dsl, const float* __restrict__ U, int G, int r, float* __restrict__ d_e2,
                           long long ld) {
    const long long n = (long long)G * r;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / r;
        const int k = (int)(t % r);
        d_e2[i * ld + k] = dsl[i] * U[t];
    }
}

__global__ void reduce_splits_kernel2(const float* __restrict__ part, int S, long long stride, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += part[(long long)s * stride + i];
        out[i] = acc;
    }
}

static inline size_t mt_align(size_t x) { return (x + 255) / 256 * 256; }

static inline int mt_splits(int M, int N, int K) { return choose_splits(M, N, K); }

}  // namespace txe

using namespace txe;

extern "C" {

// U[G][r] = E1[G][l] * W[l][r]
int txe_bilinear_project(const float* e1, long long ld_e1, int G, int l, const float* W, int r, float* U, void* stream) {
    if (G < 0 || l < 1 || r < 1 || !e1 || !W || !U) return TXE_ERR_ARG;
    VMat A = vmat_plain(e1, ld_e1, G, l);
    VMat B = vmat_plain(W, r, l, r);
    Epi E = epi_plain(U, r, r);
    return gemm_nn(A, B, E, G, r, l, 1, (hipStream_t)stream);
}

// pairwise form used in training (model.py:86): s[i] = e1_i^T W e2_i, optionally exp.  U is a G x r scratch that
// backward reuses.
int txe_bilinear_pair_fwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r,
                          const float* W, int apply_exp, float* U, float* s, void* stream) {
    if (G < 0 || !e2 || !s) return TXE_ERR_ARG;
    int rc = txe_bilinear_project(e1, ld_e1, G, l, W, r, U, stream);
    if (rc) return rc;
    if (G == 0) return TXE_OK;
    hipLaunchKernelGGL(rowdot_kernel, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)U, e2, ld_e2, G, r,
                       apply_exp, s);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

size_t txe_bilinear_pair_bwd_ws_bytes(int G, int l, int r) {
    return mt_align((size_t)(G > 0 ? G : 1) * 4) + mt_align((size_t)(G > 0 ? G : 1) * r * 4) +
           mt_align((size_t)mt_splits(l, r, G) * l * r * 4);
}

// ds: gradient of the returned scores.  d_e1 [G][l] and dW [l][r] are always written; d_e2 may be NULL.
int txe_bilinear_pair_bwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r,
                          const float* W, int apply_exp, const float* U, const float* s, const float* ds, float* d_e1,
                          long long ld_de1, float* d_e2, long long ld_de2, float* dW, void* ws, size_t ws_bytes, void* stream) {
    if (G < 0 || l < 1 || r < 1 || !e1 || !e2 || !W || !U || !s || !ds || !d_e1 || !dW || !ws) return TXE_ERR_ARG;
    if (ws_bytes < txe_bilinear_pair_bwd_ws_bytes(G, l, r)) return TXE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* dsl = (float*)ws;
    float* dU = (float*)((char*)ws + mt_align((size_t)(G > 0 ? G : 1) * 4));
    float* part = (float*)((char*)dU + mt_align((size_t)(G > 0 ? G : 1) * r * 4));
    int rc;
    if (G > 0) {
        hipLaunchKernelGGL(dsl_kernel, dim3((G + 255) / 256), dim3(256), 0, st, ds, s, apply_exp, G, dsl);
        const long long nu = (long long)G * r;
        hipLaunchKernelGGL(du_kernel, dim3((int)((nu + 255) / 256 < 2048 ? (nu + 255) / 256 : 2048)), dim3(256), 0, st,
                           (const float*)dsl, e2, ld_e2, G, r, dU);
        TXE_CHECK_LAUNCH();
        // d_e1[i][j] = sum_k dU[i][k] W[j][k]
        VMat A = vmat_plain(dU, r, G, r);
        VMat B = vmat_plain(W, r, l, r);
        Epi E = epi_plain(d_e1, ld_de1, l);
        rc = gemm_nt(A, B, E, G, l, r, 1, st);
        if (rc) return rc;
        if (d_e2) {
            const long long n = (long long)G * r;
            const int nb = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
            hipLaunchKernelGGL(de2_kernel, dim3(nb), dim3(256), 0, st, (const float*)dsl, U, G, r, d_e2, ld_de2);
            TXE_CHECK_LAUNCH();
        }
    }
    // dW[j][k] = sum_i e1[i][j] * dU[i][k]
    const int S = mt_splits(l, r, G);
    VMat A = vmat_plain(e1, ld_e1, G, l);
    VMat B = vmat_plain(dU, r, G, r);
    Epi E = epi_plain(part, r, r);
    E.split_stride = (long long)l * r;
    rc = gemm_tn(A, B, E, l, r, G, S, st);
    if (rc) return rc;
    const long long n = (long long)l * r;
    const int nb = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(reduce_splits_kernel2, dim3(nb), dim3(256), 0, st, (const float*)part, G > 0 ? S : 0, E.split_stride, n, dW);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// Plain dense product on the library's fp32 MFMA GEMM (tests / micro-benchmarks; the model paths above use the same kernels
// through their fused entry points).  layout 0: C = A[M][K] * B[N][K]^T;  1: C = A[M][K] * B[K][N];  2: C = A[K][M]^T * B[K][N].
// splits > 1 writes `splits` partial products at C + z*M*N (the caller reduces them).
size_t txe_gemm_tail_ws_bytes(void) { return gemm_tail_ws_bytes(); }

int txe_gemm_plain(int layout, const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N,
                   int K, int splits, void* ws, size_t ws_bytes, void* stream) {
    if (layout < 0 || layout > 2 || M < 0 || N < 0 || K < 0 || !A || !B || !C) return TXE_ERR_ARG;
    Epi E = epi_plain(C, ldc, N);
    if (splits > 1) E.split_stride = (long long)M * ldc;
    hipStream_t s = (hipStream_t)stream;
    if (layout == 0) return gemm_nt(vmat_plain(A, lda, M, K), vmat_plain(B, ldb, N, K), E, M, N, K, splits, s, ws, ws_bytes);
    if (layout == 1) return gemm_nn(vmat_plain(A, lda, M, K), vmat_plain(B, ldb, K, N), E, M, N, K, splits, s, ws, ws_bytes);
    return gemm_tn(vmat_plain(A, lda, K, M), vmat_plain(B, ldb, K, N), E, M, N, K, splits, s, ws, ws_bytes);
}

// One block of the scoring loop: S[q][g] = <Q[q], U[g]> (exp optionally), q < nq, g < G.  S row stride ld_s.
int txe_score_block(const float* Q, long long ld_q, int nq, const float* U, int G, int r, int apply_exp, float* S,
                    long long ld_s, void* stream) {
    if (nq < 0 || G < 0 || r < 1 || !Q || !U || !S) return TXE_ERR_ARG;
    VMat A = vmat_plain(Q, ld_q, nq, r);
    VMat B = vmat_plain(U, r, G, r);
    Epi E = epi_plain(S, ld_s, G);
    E.apply_exp = apply_exp;
    return gemm_nt(A, B, E, nq, G, r, 1, (hipStream_t)stream);
}

}  // extern "C"
