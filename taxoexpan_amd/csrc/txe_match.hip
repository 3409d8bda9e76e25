// Bilinear matchers and the all-candidate scoring loop.
//   BIM / LBM  (model_zoo.py:301-328):  s_i = e1_i^T W e2_i   (nn.Bilinear(l, r, 1, bias=False)),  LBM: exp(s_i)
//   scoring loop (test_fast.py:116-123, infer.py:95-99): for every query q, match(hg, q.expand(G,-1)) -> G scores.
// The literal loop costs 2*l*r flops per (query, candidate) pair.  Here the bilinear form is factored once,
// U = HG W  (G x r), and every query block is one NT GEMM  S[q][g] = <Q[q], U[g]>  with the exp fused into the
// epilogue -- 2*r flops per pair on the fp32 MFMA pipe.
#include <string.h>

#include "txe_gemm.h"
#include "txe_gemm_split.h"
#include "txe_skinny.h"

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

// dsl[i] = ds[i] * (apply_exp ? s[i] : 1): gradient at the bilinear form (LBM returns exp of it, model_zoo.py:325-328)
// dU[i][k] = dsl[i] * e2[i][k]      (gradient of U = E1 W, the only consumer of e2 in s_i = <U_i, e2_i>)
__global__ void du_kernel(const float* __restrict__ ds, const float* __restrict__ s, int apply_exp, const float* __restrict__ e2,
                          long long ld_e2, int G, int r, float* __restrict__ dU) {
    const long long n = (long long)G * r;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / r;
        const int k = (int)(t % r);
        const float dsl = apply_exp ? ds[i] * s[i] : ds[i];
        dU[t] = dsl * e2[i * ld_e2 + k];
    }
}

// d_e2[i][k] = dsl[i] * U[i][k]
__global__ void de2_kernel(const float* __restrict__ ds, const float* __restrict__ s, int apply_exp, const float* __restrict__ U, int G,
                           int r, float* __restrict__ d_e2, long long ld) {
    const long long n = (long long)G * r;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / r;
        const int k = (int)(t % r);
        const float dsl = apply_exp ? ds[i] * s[i] : ds[i];
        d_e2[i * ld + k] = dsl * U[t];
    }
}

// s[i] = <e1[i], V[i]> (exp optionally), one wavefront per row: the query-side form of the bilinear match, V = E2 W^T
__global__ __launch_bounds__(256) void rowdot2_kernel(const float* __restrict__ e1, long long ld_e1, const float* __restrict__ V, int G, int l,
                                                      int apply_exp, float* __restrict__ s) {
    const int w = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + w;
    if (i >= G) return;
    float acc = 0.f;
    for (int k = ln; k < l; k += 64) acc = fmaf(e1[(long long)i * ld_e1 + k], V[(long long)i * l + k], acc);
    acc = wave_sum(acc);
    if (ln == 0) s[i] = apply_exp ? __expf(acc) : acc;
}

// d_e1[i][k] = dsl[i] * V[i][k];  R[i][k] = dsl[i] * e1[i][k]   (dsl = ds * (apply_exp ? s : 1)): the whole backward of
// s_i = <e1_i, V_i> with respect to e1, and the left operand of dW = R^T E2
__global__ void bil_scale_kernel(const float* __restrict__ ds, const float* __restrict__ s, int apply_exp, const float* __restrict__ V,
                                 const float* __restrict__ e1, long long ld_e1, int G, int l, float* __restrict__ d_e1, long long ld_de1,
                                 float* __restrict__ R) {
    const long long n = (long long)G * l;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / l;
        const int k = (int)(t % l);
        const float dsl = apply_exp ? ds[i] * s[i] : ds[i];
        d_e1[i * ld_de1 + k] = dsl * V[t];
        R[t] = dsl * e1[i * ld_e1 + k];
    }
}

__global__ void reduce_splits_kernel2(const float* __restrict__ part, int S, long long stride, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int s0 = 0; s0 < S; s0 += 4) {          // slice order kept; four clamped loads in flight
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = part[(long long)min(s0 + j, S - 1) * stride + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += (s0 + j < S) ? v[j] : 0.f;
        }
        out[i] = acc;
    }
}

// y[i][o] = act(y[i][o] + b[o])      act: 0 none, 1 relu, 2 tanh
__global__ void bias_act_kernel(float* __restrict__ y, const float* __restrict__ b, long long n_rows, int O, int act) {
    const long long n = n_rows * O;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        float v = y[t] + (b ? b[t % O] : 0.f);
        if (act == 1) v = v > 0.f ? v : 0.f;
        else if (act == 2) v = tanhf(v);
        y[t] = v;
    }
}
// dz = dy * act'(y)  (in terms of the activated output y), colsum partial for the bias gradient done by the caller
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n, int act, float* __restrict__ dz) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const float yv = y[t];
        float g = 1.f;
        if (act == 1) g = yv > 0.f ? 1.f : 0.f;
        else if (act == 2) g = 1.f - yv * yv;
        dz[t] = dy[t] * g;
    }
}
__global__ void colsum_small_kernel(const float* __restrict__ x, long long n_rows, int O, float* __restrict__ out) {
    __shared__ float red[256];
    const int o = blockIdx.x;
    float a = 0.f;
    for (long long r = threadIdx.x; r < n_rows; r += blockDim.x) a += x[r * O + o];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[o] = red[0];
}

// ---- the pairwise match when the query rows repeat in RUNS (txe_bilinear_runs_*, txe_bilinear_stacked_*) ------------------------------
// The runs are either given (compact distinct rows Qu [U][r] + run offsets, U known on the host) or found on the device in the stacked
// matrix E2 [G][r] itself (run u's row is row off[u] of E2, the number of runs is a device scalar): every kernel walks the runs with a
// grid stride and reads the count through runs_count(), so one launch shape serves any count.
// flag[i] = row i of E2 differs (bit pattern) from row i - 1; flag[0] = 1.  One wave per row.
__global__ __launch_bounds__(256) void row_change_kernel(const float* __restrict__ e2, long long ld, int G, int r, int* __restrict__ flag) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
    if (i >= G) return;
    int diff = (i == 0);
    if (i > 0) {
        const unsigned* a = reinterpret_cast<const unsigned*>(e2 + (long long)i * ld);
        const unsigned* b = reinterpret_cast<const unsigned*>(e2 + (long long)(i - 1) * ld);
        for (int k = ln; k < r; k += 64) diff |= (a[k] != b[k]);
    }
    const unsigned long long any = __ballot(diff);
    if (ln == 0) flag[i] = any != 0ull;
}

// inclusive scan of the flags by ONE workgroup of 1,024 threads (chunks of 1,024 rows with a carry): run_id[i] = #flags up to i - 1, the
// first row of every run into run_off, the count into n_runs, run_off[n_runs] = G.  flag and run_id may be the same array.
__global__ __launch_bounds__(1024) void runs_scan_kernel(const int* flag, int G, int* run_id, int* __restrict__ run_off, int* __restrict__ n_runs) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, ln = t & 63, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < G; i0 += 1024) {
        const int i = i0 + t;
        const int f = i < G ? flag[i] : 0;
        int x = f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (ln >= d) x += y; }
        if (ln == 63) wsum[w] = x;
        __syncthreads();
        int before = carry_s;
        for (int q = 0; q < w; ++q) before += wsum[q];
        const int incl = before + x;
        if (i < G) {
            run_id[i] = incl - 1;
            if (f) run_off[incl - 1] = i;
        }
        __syncthreads();
        if (t == 1023) carry_s = incl;
        __syncthreads();
    }
    if (t == 0) { n_runs[0] = carry_s; run_off[carry_s] = G; }
}

// s_i = <e1_i, V[u]> for the pairs i of run u: V[u] in registers, one wave per pair, gridDim.y workgroups share a run
// one_col >= 0: column one_col of e1 counts as 1 whatever it holds (the bias row of a folded GCN layer: V[u][one_col] = <bias, ...>)
__global__ __launch_bounds__(256) void rowdot_runs_kernel(const float* __restrict__ e1, long long ld_e1, const float* __restrict__ V,
                                                          const RunsRef R, int l, int apply_exp, float* __restrict__ s, const int one_col = -1) {
    const int w = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int U = runs_count(R);
    for (int u = blockIdx.x; u < U; u += gridDim.x) {
        const int i0 = R.off[u], i1 = R.off[u + 1];
        for (int k0 = 0; k0 < l; k0 += 64 * 8) {                // (rows wider than 512 columns: several passes, partial sums in s)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int k = k0 + ln + 64 * j; v[j] = k < l ? V[(long long)u * l + k] : 0.f; }
            for (int i = i0 + w + 4 * blockIdx.y; i < i1; i += 4 * gridDim.y) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + ln + 64 * j;
                    const float x = k < l ? e1[(long long)i * ld_e1 + k] : 0.f;
                    acc = fmaf(k == one_col ? 1.f : x, v[j], acc);
                }
                acc = wave_sum(acc);
                if (ln == 0) {
                    const float tot = (k0 == 0 ? 0.f : s[i]) + acc;
                    s[i] = (apply_exp && k0 + 64 * 8 >= l) ? __expf(tot) : tot;
                }
            }
        }
    }
}

// backward of the above for run u: d_e1_i = dsl_i V[u], S[u] = sum over the run's pairs (in order) of dsl_i e1_i; thread = column
constexpr int RB_NL = 16;       // (a training run is 32 pairs: two dependent round trips instead of four)
struct RunsBwdArgs {
    const float *ds, *s; int apply_exp; const float* V; const float* e1; long long ld_e1; RunsRef R; int l; float* d_e1; long long ld_de1; float* S;
    int one_col;          // >= 0: column one_col of e1 counts as 1 (see rowdot_runs_kernel); -1: none
};
// workgroup (bx of nbx, by): the columns [by * blockDim, (by + 1) * blockDim) of the runs bx, bx + nbx, ...
__device__ __forceinline__ void runs_bwd_job(const int bx, const int nbx, const int by, const RunsBwdArgs& a) {
    const float* __restrict__ ds = a.ds; const float* __restrict__ s = a.s; const float* __restrict__ V = a.V; const float* __restrict__ e1 = a.e1;
    float* __restrict__ d_e1 = a.d_e1; float* __restrict__ S = a.S;
    const int apply_exp = a.apply_exp, l = a.l;
    const long long ld_e1 = a.ld_e1, ld_de1 = a.ld_de1;
    const RunsRef& R = a.R;
    const int k = by * (int)blockDim.x + (int)threadIdx.x;
    if (k >= l) return;
    const int U = runs_count(R);
    for (int u = bx; u < U; u += nbx) {
        const int i0 = R.off[u], i1 = R.off[u + 1];
        const float v = V[(long long)u * l + k];
        float acc = 0.f;
        for (int i = i0; i < i1; i += RB_NL) {                  // RB_NL pairs' loads in flight; the sum keeps the pairs' order
            float x[RB_NL], dsl[RB_NL];
#pragma unroll
            for (int q = 0; q < RB_NL; ++q) {
                const int ii = min(i + q, i1 - 1);
                x[q] = (k == a.one_col) ? 1.f : e1[(long long)ii * ld_e1 + k];
                dsl[q] = apply_exp ? ds[ii] * s[ii] : ds[ii];
            }
#pragma unroll
            for (int q = 0; q < RB_NL; ++q) {
                if (i + q < i1) {
                    if (d_e1) d_e1[(long long)(i + q) * ld_de1 + k] = dsl[q] * v;     // (NULL: only the run sums are wanted)
                    acc = fmaf(dsl[q], x[q], acc);
                }
            }
        }
        S[(long long)u * l + k] = acc;
    }
}
__global__ __launch_bounds__(256) void runs_bwd_kernel(const RunsBwdArgs a) { runs_bwd_job(blockIdx.x, gridDim.x, blockIdx.y, a); }

// run_id[i] = the run that holds pair i (binary search in the offsets)
__global__ void runs_expand_kernel(const int* __restrict__ off, int U, int G, int* __restrict__ run_id) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G) return;
    int lo = 0, hi = U - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= i) lo = mid; else hi = mid - 1; }
    run_id[i] = lo;
}

static inline size_t mt_align(size_t x) { return (x + 255) / 256 * 256; }

static inline int mt_splits(int M, int N, int K) { return choose_splits(M, N, K); }

// ---- the run products on the skinny MFMA kernel (txe_skinny.h) ----
// the launch shape for a run count that lives on the device: a training batch pairs a query with 1 + negative_size = 32 anchors
static inline int runs_hint(const RunsRef& R, int G) { return R.n_dev ? (G / 32 > 32 ? G / 32 : 32) : (R.n_host > 0 ? R.n_host : 1); }

// C [U][N] = A[rows of the runs] B^T (B [N][K] k-contiguous) or A B (B [K][N]); A [.][K] k-contiguous, compact rows unless a_first
static inline void skinny_runs_rows(SkinnyArgs& a, const float* A, long long lda, bool a_first, const float* B, long long ldb, bool b_kmajor,
                                    float* C, long long ldc, int N, int K, const RunsRef& R, int G) {
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.M = R.n_host; a.N = N; a.K = K; a.m_dyn = 1; a.a_rows_first = a_first ? 1 : 0; a.R = R;
    skinny_setup(a, false, b_kmajor, runs_hint(R, G), K);
}
// C [M][N] = sum over the runs u of A[u][m] B[row(u)][n]   (both operands one row per run; B's rows through the runs when b_first)
static inline void skinny_runs_sum(SkinnyArgs& a, const float* A, long long lda, const float* B, long long ldb, bool b_first, float* C,
                                   long long ldc, int M, int N, const RunsRef& R, int G) {
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = R.n_host; a.k_dyn = 1; a.b_k_first = b_first ? 1 : 0; a.R = R;
    skinny_setup(a, true, true, M, runs_hint(R, G));
}
static int skinny_one(const char* name, SkinnyArgs& a, double bytes, hipStream_t st) {
    SkinnyMulti m;
    memset(&m, 0, sizeof(m));
    m.n = 1; m.j[0] = a;
    ProfScope prof(name, st, bytes, 1);
    return skinny_launch(m, st);
}

}  // namespace txe

using namespace txe;

extern "C" {

// U[G][r] = E1[G][l] * W[l][r]   (row stride ld_u >= r: a multiple of 4 lets the scoring GEMM read U with 16-byte loads)
// sws (or NULL: the fp32 MFMA): txe_gemm_plain_split_ws_bytes(G, r, l) of scratch -- the product on the bf16 matrix pipe in fp32 accuracy
// (txe_gemm_split.h; W packed from its transpose)
int txe_bilinear_project(const float* e1, long long ld_e1, int G, int l, const float* W, int r, float* U, long long ld_u,
                         void* sws, size_t sws_bytes, void* stream) {
    if (G < 0 || l < 1 || r < 1 || ld_u < r || !e1 || !W || !U) return TXE_ERR_ARG;
    if (sws && G > 0) {
        if (sws_bytes < split_pair_bytes(G, r, l)) return TXE_ERR_WORKSPACE;
        char* w = (char*)sws;
        const size_t a = (split_packed_bytes(G, l) + 255) / 256 * 256;
        int rc = split_pack_launch(e1, ld_e1, G, l, 0, w, (hipStream_t)stream);
        if (rc) return rc;
        rc = split_pack_launch(W, r, r, l, 3, w + a, (hipStream_t)stream);      // W [l][r] = the transpose of the column operand [r][l]
        if (rc) return rc;
        return gemm_nt_split_launch(w, w + a, G, r, l, U, ld_u, 0.0, (hipStream_t)stream);
    }
    VMat A = vmat_plain(e1, ld_e1, G, l);
    VMat B = vmat_plain(W, r, l, r);
    Epi E = epi_plain(U, ld_u, r);
    return gemm_nn(A, B, E, G, r, l, 1, (hipStream_t)stream);
}

// pairwise form used in training (model.py:86): s[i] = e1_i^T W e2_i, optionally exp.  U is a G x r scratch that
// backward reuses.
int txe_bilinear_pair_fwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r,
                          const float* W, int apply_exp, float* U, float* s, void* stream) {
    if (G < 0 || !e2 || !s) return TXE_ERR_ARG;
    int rc = txe_bilinear_project(e1, ld_e1, G, l, W, r, U, r, nullptr, 0, stream);
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
    float* dU = (float*)((char*)ws + mt_align((size_t)(G > 0 ? G : 1) * 4));
    float* part = (float*)((char*)dU + mt_align((size_t)(G > 0 ? G : 1) * r * 4));
    int rc;
    if (G > 0) {
        const long long nu = (long long)G * r;
        hipLaunchKernelGGL(du_kernel, dim3((int)((nu + 255) / 256 < 2048 ? (nu + 255) / 256 : 2048)), dim3(256), 0, st, ds, s, apply_exp,
                           e2, ld_e2, G, r, dU);
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
            hipLaunchKernelGGL(de2_kernel, dim3(nb), dim3(256), 0, st, ds, s, apply_exp, U, G, r, d_e2, ld_de2);
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

// Query-side form of the pairwise match when e2 (the query features) needs no gradient -- always, in training (model.py:86, trainer.py:51):
//   forward   V = E2 W^T [G][l] (one GEMM);  s_i = <e1_i, V_i>
//   backward  d_e1_i = dsl_i V_i (elementwise -- the candidate-side form needs a second G-row GEMM here);  dW = (dsl (.) E1)^T E2
// V [G][l] is kept for backward.
int txe_bilinear_query_project(const float* e2, long long ld_e2, int G, int l, int r, const float* W, float* V, void* stream) {
    if (G < 0 || l < 1 || r < 1 || !e2 || !W || !V) return TXE_ERR_ARG;
    if (G == 0) return TXE_OK;
    VMat A = vmat_plain(e2, ld_e2, G, r);
    VMat B = vmat_plain(W, r, l, r);
    Epi E = epi_plain(V, l, l);
    return gemm_nt(A, B, E, G, l, r, 1, (hipStream_t)stream);
}

int txe_bilinear_query_dot(const float* e1, long long ld_e1, const float* V, int G, int l, int apply_exp, float* s, void* stream) {
    if (G < 0 || l < 1 || !e1 || !V || !s) return TXE_ERR_ARG;
    if (G == 0) return TXE_OK;
    hipLaunchKernelGGL(rowdot2_kernel, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, e1, ld_e1, V, G, l, apply_exp, s);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_bilinear_query_fwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r, const float* W,
                           int apply_exp, float* V, float* s, void* stream) {
    if (!e1 || !s) return TXE_ERR_ARG;
    int rc = txe_bilinear_query_project(e2, ld_e2, G, l, r, W, V, stream);
    if (rc) return rc;
    return txe_bilinear_query_dot(e1, ld_e1, V, G, l, apply_exp, s, stream);
}

size_t txe_bilinear_query_bwd_ws_bytes(int G, int l, int r) {
    return mt_align((size_t)(G > 0 ? G : 1) * l * 4) + mt_align((size_t)mt_splits(l, r, G) * l * r * 4);
}

int txe_bilinear_query_bwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r, int apply_exp,
                           const float* V, const float* s, const float* ds, float* d_e1, long long ld_de1, float* dW, void* ws,
                           size_t ws_bytes, void* stream) {
    if (G < 0 || l < 1 || r < 1 || !e1 || !e2 || !V || !s || !ds || !d_e1 || !dW || !ws) return TXE_ERR_ARG;
    if (ws_bytes < txe_bilinear_query_bwd_ws_bytes(G, l, r)) return TXE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* R = (float*)ws;
    float* part = (float*)((char*)ws + mt_align((size_t)(G > 0 ? G : 1) * l * 4));
    if (G > 0) {
        const long long n = (long long)G * l;
        hipLaunchKernelGGL(bil_scale_kernel, dim3((int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, st, ds, s, apply_exp, V, e1,
                           ld_e1, G, l, d_e1, ld_de1, R);
        TXE_CHECK_LAUNCH();
    }
    // dW[j][k] = sum_i R[i][j] * e2[i][k]
    const int S = mt_splits(l, r, G);
    VMat A = vmat_plain(R, l, G, l);
    VMat B = vmat_plain(e2, ld_e2, G, r);
    Epi E = epi_plain(part, r, r);
    E.split_stride = (long long)l * r;
    int rc = gemm_tn(A, B, E, l, r, G, S, st);
    if (rc) return rc;
    const long long n = (long long)l * r;
    hipLaunchKernelGGL(reduce_splits_kernel2, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st, (const float*)part,
                       G > 0 ? S : 0, E.split_stride, n, dW);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// The same match when the query rows REPEAT IN RUNS -- a training batch pairs one query with 1 + negative_size consecutive anchors
// and the reference's collate stacks that query's row for each of them (data_loaders.py:9-28).  Qu [U][r]: the distinct rows;
// run_off [U+1]: the first pair of every run, run_off[U] = G.  V = Qu W^T is U rows instead of G, backward's dW = S^T Qu with
// S[u] = sum over run u of dsl_i e1_i (pairs in order: deterministic) has K = U instead of G; d_e1_i = dsl_i V[u].  Both U-row products
// run on small dot-product kernels (a few hundred rows would leave the MFMA GEMM's 128-row tiles most of the chip idle: 41 + 29 us measured).
static int runs_fwd_launch(const float* e1, long long ld_e1, const float* Q, long long ld_q, const RunsRef& R, int gx, int G, int l, int r,
                           const float* W, int apply_exp, float* V, float* s, hipStream_t st) {
    {   // V [U][l] = Q[run rows] W^T
        SkinnyArgs a;
        skinny_runs_rows(a, Q, ld_q, R.first_row != 0, W, (long long)r, false, V, (long long)l, l, r, R, G);
        const int rc = skinny_one("skinny_gemm_kernel[V]", a, 4.0 * ((double)runs_hint(R, G) * (r + l) + (double)l * r), st);
        if (rc) return rc;
    }
    {
        ProfScope prof("rowdot_runs_kernel", st, 4.0 * ((double)G * l + (double)R.n_host * l + G), 1);
        hipLaunchKernelGGL(rowdot_runs_kernel, dim3(gx, 8), dim3(256), 0, st, e1, ld_e1, (const float*)V, R, l, apply_exp, s);
    }
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

static int runs_bwd_launch(const float* e1, long long ld_e1, const float* Q, long long ld_q, const RunsRef& R, int gx, int G, int l, int r,
                           int apply_exp, const float* V, const float* s, const float* ds, float* d_e1, long long ld_de1, float* dW, float* S,
                           hipStream_t st) {
    {
        ProfScope prof("runs_bwd_kernel", st, 4.0 * (2.0 * G * l + 2.0 * R.n_host * l + 2.0 * G), 1);
        const RunsBwdArgs ba{ds, s, apply_exp, V, e1, ld_e1, R, l, d_e1, ld_de1, S, -1};
        hipLaunchKernelGGL(runs_bwd_kernel, dim3(gx, (l + 255) / 256), dim3(256), 0, st, ba);
    }
    TXE_CHECK_LAUNCH();
    {   // dW [l][r] = sum_u S[u]^T q_u
        SkinnyArgs a;
        skinny_runs_sum(a, S, (long long)l, Q, ld_q, R.first_row != 0, dW, (long long)r, l, r, R, G);
        const int rc = skinny_one("skinny_gemm_kernel[dWm]", a, 4.0 * ((double)runs_hint(R, G) * (r + l) + (double)l * r), st);
        if (rc) return rc;
    }
    return TXE_OK;
}

int txe_bilinear_runs_fwd(const float* e1, long long ld_e1, const float* Qu, long long ld_q, const int* run_off, int G, int U, int l, int r,
                          const float* W, int apply_exp, float* V, float* s, void* stream) {
    if (G < 0 || U < 0 || l < 1 || r < 1 || !e1 || !Qu || !run_off || !W || !V || !s) return TXE_ERR_ARG;
    if (G == 0 || U == 0) return TXE_OK;
    const RunsRef R{run_off, nullptr, U, 0};
    return runs_fwd_launch(e1, ld_e1, Qu, ld_q, R, U, G, l, r, W, apply_exp, V, s, (hipStream_t)stream);
}

size_t txe_bilinear_runs_bwd_ws_bytes(int U, int l, int r) { return mt_align((size_t)(U > 0 ? U : 1) * l * 4); }

int txe_bilinear_runs_bwd(const float* e1, long long ld_e1, const float* Qu, long long ld_q, const int* run_off, int G, int U, int l, int r,
                          int apply_exp, const float* V, const float* s, const float* ds, float* d_e1, long long ld_de1, float* dW, void* ws,
                          size_t ws_bytes, void* stream) {
    if (G < 0 || U < 0 || l < 1 || r < 1 || !e1 || !Qu || !run_off || !V || !s || !ds || !d_e1 || !dW || !ws) return TXE_ERR_ARG;
    if (ws_bytes < txe_bilinear_runs_bwd_ws_bytes(U, l, r)) return TXE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (G == 0 || U == 0) {                                    // an empty sum
        if (hipMemsetAsync(dW, 0, (size_t)l * r * sizeof(float), st) != hipSuccess) return TXE_ERR_LAUNCH;
        return TXE_OK;
    }
    const RunsRef R{run_off, nullptr, U, 0};
    return runs_bwd_launch(e1, ld_e1, Qu, ld_q, R, U, G, l, r, apply_exp, V, s, ds, d_e1, ld_de1, dW, (float*)ws, st);
}

// The runs found ON THE DEVICE in the stacked query matrix the reference's collate hands over (E2 [G][r], one row per pair): rows are
// compared bit for bit with their predecessor, a one-workgroup scan numbers the runs.  run_id [G], run_off [G + 1], n_runs [1] (device).
int txe_rows_find_runs(const float* e2, long long ld_e2, int G, int r, int* run_id, int* run_off, int* n_runs, void* stream) {
    if (G < 0 || r < 1 || !e2 || !run_id || !run_off || !n_runs) return TXE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (G > 0) {
        ProfScope prof("row_change_kernel", st, 4.0 * ((double)G * r + G), 1);
        hipLaunchKernelGGL(row_change_kernel, dim3((G + 3) / 4), dim3(256), 0, st, e2, ld_e2, G, r, run_id);
    }
    TXE_CHECK_LAUNCH();
    {
        ProfScope prof("runs_scan_kernel", st, 4.0 * 3.0 * G, 1);
        hipLaunchKernelGGL(runs_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)run_id, G, run_id, run_off, n_runs);
    }
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// txe_bilinear_runs_* on the stacked matrix with the runs of txe_rows_find_runs (count on the device: V and the workspace are sized for
// the worst case, G runs; the kernels walk the runs with a grid stride).  Same values as txe_bilinear_query_* up to the summation order.
int txe_bilinear_stacked_fwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, const int* run_off, const int* n_runs, int G,
                             int l, int r, const float* W, int apply_exp, float* V, float* s, void* stream) {
    if (G < 0 || l < 1 || r < 1 || !e1 || !e2 || !run_off || !n_runs || !W || !V || !s) return TXE_ERR_ARG;
    if (G == 0) return TXE_OK;
    const RunsRef R{run_off, n_runs, G, 1};
    const int gx = G < 512 ? G : 512;
    return runs_fwd_launch(e1, ld_e1, e2, ld_e2, R, gx, G, l, r, W, apply_exp, V, s, (hipStream_t)stream);
}

size_t txe_bilinear_stacked_bwd_ws_bytes(int G, int l, int r) { return mt_align((size_t)(G > 0 ? G : 1) * l * 4); }

int txe_bilinear_stacked_bwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, const int* run_off, const int* n_runs, int G,
                             int l, int r, int apply_exp, const float* V, const float* s, const float* ds, float* d_e1, long long ld_de1,
                             float* dW, void* ws, size_t ws_bytes, void* stream) {
    if (G < 0 || l < 1 || r < 1 || !e1 || !e2 || !run_off || !n_runs || !V || !s || !ds || !d_e1 || !dW || !ws) return TXE_ERR_ARG;
    if (ws_bytes < txe_bilinear_stacked_bwd_ws_bytes(G, l, r)) return TXE_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (G == 0) {
        if (hipMemsetAsync(dW, 0, (size_t)l * r * sizeof(float), st) != hipSuccess) return TXE_ERR_LAUNCH;
        return TXE_OK;
    }
    const RunsRef R{run_off, n_runs, G, 1};
    const int gx = G < 512 ? G : 512;
    return runs_bwd_launch(e1, ld_e1, e2, ld_e2, R, gx, G, l, r, apply_exp, V, s, ds, d_e1, ld_de1, dW, (float*)ws, st);
}

// graph -> run index for runs given by their offsets (txe_rows_find_runs produces the same array for the stacked form)
int txe_runs_expand(const int* run_off, int U, int G, int* run_id, void* stream) {
    if (U < 0 || G < 0 || !run_off || !run_id) return TXE_ERR_ARG;
    if (U == 0 || G == 0) return TXE_OK;
    hipLaunchKernelGGL(runs_expand_kernel, dim3((G + 255) / 256), dim3(256), 0, (hipStream_t)stream, run_off, U, G, run_id);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// The pairwise bilinear match FOLDED through the output layer of the encoder (the graph vector hg = Z Wf^T of txe_gat_collapse_fwd, never
// formed):  s_i = hg_i^T Wm q_i = <Z_i, T[u(i)]>,  T[u] = Wf^T (Wm q_u)  -- when the query rows repeat in runs, the D x Kp product runs
// on U run rows instead of G graph rows (a training batch: 128 instead of 4,096), forward and both backward products.
//   Z [G][Kp] (row pitch ld_z), Wf [l][Kp] (the packed weight rows, pitch ld_wf), Wm [l][r];  runs as in txe_bilinear_runs_* (first_row 0:
//   Q = the U distinct rows, n_runs NULL) or txe_bilinear_stacked_* (first_row 1: Q = the stacked matrix, n_runs on the device, U = G bounds
//   the buffers).  V [U][l], T [U][Kp] are kept for backward.
int txe_bilinear_folded_fwd(const float* Z, long long ld_z, int G, int Kp, const float* Wf, long long ld_wf, int l, const float* Q, long long ld_q,
                            int r, const int* run_off, const int* n_runs, int U, int first_row, const float* Wm, int apply_exp, float* V, float* T,
                            float* s, int stages, int wf_by_k, int one_col, void* stream) {
    // wf_by_k: 0 = Wf [l][Kp] (a GAT layer's packed weight rows), 1 = Wf [Kp][ld_wf] (a GCN layer's packing: row k, l columns -- row one_col
    // holds the bias, and column one_col of Z counts as 1: hg = Z Wf + b without a bias pass)
    // stages: 1 = V and T (need the queries and the weights only: the encoder asks for T before its Z sweep), 2 = the scores (needs Z), 3 = both
    if (G < 0 || U < 0 || Kp < 1 || l < 1 || r < 1 || !Wf || !Q || !run_off || !Wm || !V || !T || (first_row && !n_runs) || !(stages & 3) ||
        ((stages & 2) && (!Z || !s)) || one_col >= Kp)
        return TXE_ERR_ARG;
    if (G == 0 || U == 0) return TXE_OK;
    hipStream_t st = (hipStream_t)stream;
    const RunsRef R{run_off, n_runs, U, first_row ? 1 : 0};
    const RunsRef Rc{run_off, n_runs, U, 0};                       // the same runs, compact rows (V, T)
    const int gx = n_runs ? (G < 512 ? G : 512) : U;
    const double uh = (double)runs_hint(R, G);
    if (stages & 1) {   // V [U][l] = Q[run rows] Wm^T
        SkinnyArgs a;
        skinny_runs_rows(a, Q, ld_q, first_row != 0, Wm, (long long)r, false, V, (long long)l, l, r, R, G);
        const int rc = skinny_one("skinny_gemm_kernel[V]", a, 4.0 * (uh * (r + l) + (double)l * r), st);
        if (rc) return rc;
    }
    if (stages & 1) {   // T [U][Kp] = V Wf   (wf_by_k: V Wf^T)
        SkinnyArgs a;
        skinny_runs_rows(a, V, (long long)l, false, Wf, ld_wf, wf_by_k == 0, T, (long long)Kp, Kp, l, Rc, G);
        const int rc = skinny_one("skinny_gemm_kernel[T]", a, 4.0 * (uh * (l + Kp) + (double)l * Kp), st);
        if (rc) return rc;
    }
    if (stages & 2) {
        ProfScope prof("rowdot_runs_kernel", st, 4.0 * ((double)G * Kp + (double)U * Kp + G), 1);
        hipLaunchKernelGGL(rowdot_runs_kernel, dim3(gx, 8), dim3(256), 0, st, Z, ld_z, (const float*)T, Rc, Kp, apply_exp, s, one_col);
    }
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

//   dZ_i = dsl_i T[u(i)],  dT[u] = sum_{i in u} dsl_i Z_i,  dV = dT Wf^T,  dWf = V^T dT [l][Kp] (the main part of the output layer's weight
//   gradient: txe_gat_collapse_bwd_fused adds the attention rows' part),  dWm = dV^T Q.  dT [U][Kp], dV [U][l]: scratch.
int txe_bilinear_folded_bwd(const float* Z, long long ld_z, int G, int Kp, const float* Wf, long long ld_wf, int l, const float* Q, long long ld_q,
                            int r, const int* run_off, const int* n_runs, int U, int first_row, int apply_exp, const float* V, const float* T,
                            const float* s, const float* ds, float* dZ, long long ld_dz, float* dT, float* dV, float* dWm, float* dWf, int wf_by_k,
                            int one_col, void* stream) {
    // wf_by_k / one_col as in txe_bilinear_folded_fwd; dWf then is [Kp][l] (row one_col = the bias gradient)
    if (G < 0 || U < 0 || Kp < 1 || l < 1 || r < 1 || !Z || !Wf || !Q || !run_off || !V || !T || !s || !ds || !dT || !dV || !dWm || !dWf ||
        (first_row && !n_runs) || (wf_by_k ? ld_wf < l : ld_wf != Kp) || one_col >= Kp)
        return TXE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (G == 0 || U == 0) {
        if (hipMemsetAsync(dWm, 0, (size_t)l * r * sizeof(float), st) != hipSuccess) return TXE_ERR_LAUNCH;
        if (hipMemsetAsync(dWf, 0, (size_t)l * Kp * sizeof(float), st) != hipSuccess) return TXE_ERR_LAUNCH;
        return TXE_OK;
    }
    const RunsRef R{run_off, n_runs, U, first_row ? 1 : 0};
    const RunsRef Rc{run_off, n_runs, U, 0};
    const int gx = n_runs ? (G < 512 ? G : 512) : U;
    const double uh = (double)runs_hint(R, G);
    const RunsBwdArgs ba{ds, s, apply_exp, T, Z, ld_z, Rc, Kp, dZ, ld_dz, dT, one_col};
    {
        ProfScope prof("runs_bwd_kernel", st, 4.0 * (2.0 * G * Kp + 2.0 * U * Kp + 2.0 * G), 1);
        hipLaunchKernelGGL(runs_bwd_kernel, dim3(gx, (Kp + 255) / 256), dim3(256), 0, st, ba);
    }
    {   // dV [U][l] = dT Wf^T and dWf [l][Kp] = V^T dT need dT only, not each other: one launch
        SkinnyMulti m;
        memset(&m, 0, sizeof(m));
        m.n = 2;
        skinny_runs_rows(m.j[0], dT, (long long)Kp, false, Wf, ld_wf, wf_by_k != 0, dV, (long long)l, l, Kp, Rc, G);
        if (wf_by_k) skinny_runs_sum(m.j[1], dT, (long long)Kp, V, (long long)l, false, dWf, (long long)l, Kp, l, Rc, G);     // dWf [Kp][l] = dT^T V
        else skinny_runs_sum(m.j[1], V, (long long)l, dT, (long long)Kp, false, dWf, (long long)Kp, l, Kp, Rc, G);           // dWf [l][Kp] = V^T dT
        ProfScope prof("skinny_gemm_kernel[dV+dWf]", st, 4.0 * (2.0 * uh * Kp + 2.0 * l * Kp + 2.0 * uh * l), 1);
        const int rc = skinny_launch(m, st);
        if (rc) return rc;
    }
    {   // dWm [l][r] = sum_u dV[u]^T q_u
        SkinnyArgs a;
        skinny_runs_sum(a, dV, (long long)l, Q, ld_q, first_row != 0, dWm, (long long)r, l, r, R, G);
        const int rc = skinny_one("skinny_gemm_kernel[dWm]", a, 4.0 * (uh * (l + r) + (double)l * r), st);
        if (rc) return rc;
    }
    return TXE_OK;
}

int txe_topk_merge(const float* keys, const int* idx, int nq, long long cnt, int k, int idx_base, int* out_idx, float* out_key, void* stream);

// Plain dense product on the library's fp32 MFMA GEMM (tests / micro-benchmarks; the model paths above use the same kernels
// through their fused entry points).  layout 0: C = A[M][K] * B[N][K]^T;  1: C = A[M][K] * B[K][N];  2: C = A[K][M]^T * B[K][N].
// splits > 1 writes `splits` partial products at C + z*M*N (the caller reduces them).
// route: 0 = the route the model paths take; test bits (bit-equal alternatives the parity tests compare): 1 whole rounds on gemm_kernel
// instead of the persistent kernel, 2 split-K TN products without the LDS-direct copies, 4 every eligible split-K product on 128 x 160 tiles.
size_t txe_gemm_tail_ws_bytes(void) { return gemm_tail_ws_bytes(); }

// route bit 8 (layout 0, splits 1): the product on the bf16 matrix pipe in fp32 accuracy (txe_gemm_split.h); ws then holds the packed
// operands: txe_gemm_plain_split_ws_bytes(M, N, K)
size_t txe_gemm_plain_split_ws_bytes(int M, int N, int K) { return (M < 1 || N < 1 || K < 1) ? 0 : split_pair_bytes(M, N, K); }

int txe_gemm_plain(int layout, const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N,
                   int K, int splits, int route, void* ws, size_t ws_bytes, void* stream) {
    if (layout < 0 || layout > 2 || M < 0 || N < 0 || K < 0 || !A || !B || !C || route < 0 || route > 15) return TXE_ERR_ARG;
    if (route & 8) {
        if (layout != 0 || splits > 1) return TXE_ERR_ARG;
        if (M == 0 || N == 0 || K == 0) return TXE_OK;
        if (!ws || ws_bytes < split_pair_bytes(M, N, K)) return TXE_ERR_WORKSPACE;
        char* w = (char*)ws;
        const size_t a = (split_packed_bytes(M, K) + 255) / 256 * 256;
        int rc = split_pack_launch(A, lda, M, K, 0, w, (hipStream_t)stream);
        if (rc) return rc;
        rc = split_pack_launch(B, ldb, N, K, 1, w + a, (hipStream_t)stream);
        if (rc) return rc;
        return gemm_nt_split_launch(w, w + a, M, N, K, C, ldc, 0.0, (hipStream_t)stream);
    }
    Epi E = epi_plain(C, ldc, N);
    E.route = route;
    if (splits > 1) E.split_stride = (long long)M * ldc;
    hipStream_t s = (hipStream_t)stream;
    if (layout == 0) return gemm_nt(vmat_plain(A, lda, M, K), vmat_plain(B, ldb, N, K), E, M, N, K, splits, s, ws, ws_bytes);
    if (layout == 1) return gemm_nn(vmat_plain(A, lda, M, K), vmat_plain(B, ldb, K, N), E, M, N, K, splits, s, ws, ws_bytes);
    return gemm_tn(vmat_plain(A, lda, K, M), vmat_plain(B, ldb, K, N), E, M, N, K, splits, s, ws, ws_bytes);
}

// nn.Linear on the concatenation of up to two inputs, with activation (the MLP matcher, model_zoo.py:285-298):
//   y[G][O] = act([x1 | x2] W^T + b),  W [O][l+r], x2/b optional, act 0 none / 1 relu / 2 tanh.  The concat is virtual (VMat).
int txe_linear_fwd(const float* x1, long long ld1, int l, const float* x2, long long ld2, int r, int G, const float* W, const float* b,
                   int O, int act, float* y, void* stream) {
    if (G < 0 || l < 1 || r < 0 || O < 1 || !x1 || !W || !y || (r > 0 && !x2)) return TXE_ERR_ARG;
    if (G == 0) return TXE_OK;
    hipStream_t s = (hipStream_t)stream;
    VMat A = vmat_plain(x1, ld1, G, l + r);
    A.cols_main = l; A.p2 = x2; A.ld2 = ld2;
    VMat B = vmat_plain(W, l + r, O, l + r);
    Epi E = epi_plain(y, O, O);
    int rc = gemm_nt(A, B, E, G, O, l + r, 1, s);
    if (rc) return rc;
    if (b || act) {
        const long long n = (long long)G * O;
        hipLaunchKernelGGL(bias_act_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, y, b, (long long)G, O, act);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

size_t txe_linear_bwd_ws_bytes(int G, int l, int r, int O) {
    return mt_align((size_t)(G > 0 ? G : 1) * O * 4) + mt_align((size_t)choose_splits(O, l + r, G) * O * (l + r) * 4);
}

// backward of txe_linear_fwd given dy and the activated output y: dx1 [G][l], dx2 [G][r] (NULL to skip), dW [O][l+r], db [O] (NULL ok)
int txe_linear_bwd(const float* x1, long long ld1, int l, const float* x2, long long ld2, int r, int G, const float* W, int O, int act,
                   const float* y, const float* dy, float* dx1, long long ld_dx1, float* dx2, long long ld_dx2, float* dW, float* db,
                   void* ws, size_t ws_bytes, void* stream) {
    if (G < 0 || l < 1 || r < 0 || O < 1 || !x1 || !W || !y || !dy || !dW || !ws || (r > 0 && !x2)) return TXE_ERR_ARG;
    if (ws_bytes < txe_linear_bwd_ws_bytes(G, l, r, O)) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* dz = (float*)ws;
    float* part = (float*)((char*)ws + mt_align((size_t)(G > 0 ? G : 1) * O * 4));
    const int K = l + r;
    int rc;
    if (G > 0) {
        const long long n = (long long)G * O;
        hipLaunchKernelGGL(act_bwd_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, dy, y, n, act, dz);
        TXE_CHECK_LAUNCH();
        if (db) {
            hipLaunchKernelGGL(colsum_small_kernel, dim3(O), dim3(256), 0, s, (const float*)dz, (long long)G, O, db);
            TXE_CHECK_LAUNCH();
        }
        if (dx1) {   // dx = dz W  -> split into the two inputs by the epilogue
            VMat A = vmat_plain(dz, O, G, O);
            VMat B = vmat_plain(W, K, O, K);
            Epi E = epi_plain(dx1, ld_dx1, dx2 ? l : K);
            if (dx2) { E.c2 = dx2; E.ldc2 = ld_dx2; }
            rc = gemm_nn(A, B, E, G, dx2 ? K : l, O, 1, s);
            if (rc) return rc;
        }
    }
    const int S = choose_splits(O, K, G);
    VMat A = vmat_plain(dz, O, G, O);
    VMat B = vmat_plain(x1, ld1, G, K);
    B.cols_main = l; B.p2 = x2; B.ld2 = ld2;
    Epi E = epi_plain(part, K, K);
    E.split_stride = (long long)O * K;
    rc = gemm_tn(A, B, E, O, K, G, S, s);
    if (rc) return rc;
    const long long n = (long long)O * K;
    hipLaunchKernelGGL(reduce_splits_kernel2, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, (const float*)part,
                       G > 0 ? S : 0, E.split_stride, n, dW);
    TXE_CHECK_LAUNCH();
    if (db && G == 0) (void)hipMemsetAsync(db, 0, (size_t)O * 4, s);
    return TXE_OK;
}


// The scoring loop on the bf16 matrix pipe (txe_gemm_split.h; fp32-accurate): Q and U packed into `sws` (txe_score_split_ws_bytes), then
// the NT split product through gemm_kernel's own epilogue.  All four entry points take the route together or not at all -- the fused
// ranking compares scores of different launches bit for bit, and a score is a function of its two rows' values only on either route.
}  // extern "C"
namespace txe {
static inline size_t score_split_bytes(int nq, int G, int r) {
    const size_t a = (split_packed_bytes(nq, r) + 255) / 256 * 256, b = (split_packed_bytes(G, r) + 255) / 256 * 256;
    return a + b;
}
// 1 = done on the split route, 0 = not taken (no workspace), < 0 = error
// u_packed: the candidates' planes (txe_split_pack(U, ld_u, G, r, side 1)) made ONCE per candidate set by the caller -- the loop over query
// blocks then packs only its queries (a MAG-Full block packed 620 MB of U again for every 1,024 queries: 10.8 % of the inference profile)
static int score_split(const float* Q, long long ld_q, int nq, const float* U, long long ld_u, int G, int r, const Epi& E, void* sws,
                       size_t sws_bytes, const void* u_packed, hipStream_t s) {
    if (!sws || nq < 1 || G < 1) return 0;
    const size_t a = (split_packed_bytes(nq, r) + 255) / 256 * 256;
    if (sws_bytes < (u_packed ? a : score_split_bytes(nq, G, r))) return TXE_ERR_WORKSPACE;
    char* w = (char*)sws;
    int rc = split_pack_launch(Q, ld_q, nq, r, 0, w, s);
    if (rc) return rc;
    if (!u_packed) {
        rc = split_pack_launch(U, ld_u, G, r, 1, w + a, s);
        if (rc) return rc;
    }
    rc = gemm_nt_split_epi_launch(w, u_packed ? u_packed : (const void*)(w + a), E, nq, G, r, s);
    return rc ? rc : 1;
}
}  // namespace txe
extern "C" {
size_t txe_score_split_ws_bytes(int nq, int G, int r) { return (nq < 1 || G < 1 || r < 1) ? 0 : score_split_bytes(nq, G, r); }

// One block of the scoring loop: S[q][g] = <Q[q], U[g]> (exp optionally), q < nq, g < G.  S row stride ld_s.
int txe_score_block(const float* Q, long long ld_q, int nq, const float* U, long long ld_u, int G, int r, int apply_exp, float* S,
                    long long ld_s, void* ws, size_t ws_bytes, void* sws, size_t sws_bytes, const void* u_packed, void* stream) {
    if (nq < 0 || G < 0 || r < 1 || ld_u < r || !Q || !U || !S) return TXE_ERR_ARG;
    VMat A = vmat_plain(Q, ld_q, nq, r);
    VMat B = vmat_plain(U, ld_u, G, r);
    Epi E = epi_plain(S, ld_s, G);
    E.apply_exp = apply_exp;
    // with the GEMM scratch (txe_gemm_tail_ws_bytes) the whole rounds of score tiles run on persistent workgroups (short reduction,
    // 11.7 GB of scores per 8,192 MAG-Full queries: the C stores are the cost).  Never split along k: the fused ranking compares
    // scores of different launches (thresholds from here, the count epilogue's own tiles) bit for bit
    E.plain_k_order = 1;
    {
        const int sr = score_split(Q, ld_q, nq, U, ld_u, G, r, E, sws, sws_bytes, u_packed, (hipStream_t)stream);
        if (sr != 0) return sr < 0 ? sr : TXE_OK;
    }
    const bool ok = ws && ws_bytes >= gemm_tail_ws_bytes();
    return gemm_nt(A, B, E, nq, G, r, 1, (hipStream_t)stream, ok ? ws : nullptr, ok ? ws_bytes : 0);
}

// The thresholds of the fused ranking: thr[j] = match(Q[q], Up[j]) for j in [pos_off[q], pos_off[q+1]) -- Up [n_pos][r] holds the
// candidate rows of the queries' true parents, query by query (gathered by the caller).  The same MFMA kernel and k order as
// txe_score_block (bit-identical values), but only the tiles along the staircase {(q, j) : pos_off[q] <= j < pos_off[q+1]} are
// computed and only those pairs are stored: ~(nq/128 + n_pos/128) tiles where the [nq x n_pos] product has their product.
int txe_score_positives(const float* Q, long long ld_q, int nq, const float* Up, long long ld_u, int n_pos, int r, int apply_exp,
                        const int* pos_off, float* thr, void* sws, size_t sws_bytes, void* stream) {
    if (nq < 0 || n_pos < 0 || r < 1 || ld_u < r || !Q || !Up || !pos_off || !thr) return TXE_ERR_ARG;
    if (nq == 0 || n_pos == 0) return TXE_OK;
    VMat A = vmat_plain(Q, ld_q, nq, r);
    VMat B = vmat_plain(Up, ld_u, n_pos, r);
    Epi E = epi_plain(thr, 0, n_pos);
    E.apply_exp = apply_exp;
    E.cnt_mode = 3;
    E.cnt_off = pos_off;
    E.plain_k_order = 1;
    E.alg_flops = 2.0 * n_pos * (double)r;           // the pairs that are wanted
    {
        const int sr = score_split(Q, ld_q, nq, Up, ld_u, n_pos, r, E, sws, sws_bytes, nullptr, (hipStream_t)stream);
        if (sr != 0) return sr < 0 ? sr : TXE_OK;
    }
    return gemm_nt(A, B, E, nq, n_pos, r, 1, (hipStream_t)stream);
}

// Fused scoring + ranking of one block of the loop: the score tile never leaves the workgroup.
//   counts[j] += #{ g < G : match(Q[q], U[g]) strictly better than thr[j] }   for j in [pos_off[q], pos_off[q+1])
// thr[j] = the score of query q's j-th true parent, computed by THIS library's score kernel (txe_score_block on the gathered
// rows: identical k-order, hence bit-identical values); counts are int32, zeroed by the caller, accumulated with atomics --
// exact and order independent.  Candidates may be a shard: counts of shards add.  txe_rank_finalize turns them into ranks.
int txe_score_count_block(const float* Q, long long ld_q, int nq, const float* U, long long ld_u, int G, int r, int apply_exp,
                          const int* pos_off, const float* thr, int larger_is_better, int* counts, void* sws, size_t sws_bytes,
                          const void* u_packed, void* stream) {
    if (nq < 0 || G < 0 || r < 1 || ld_u < r || !Q || !U || !pos_off || !thr || !counts) return TXE_ERR_ARG;
    if (nq == 0 || G == 0) return TXE_OK;
    VMat A = vmat_plain(Q, ld_q, nq, r);
    VMat B = vmat_plain(U, ld_u, G, r);
    Epi E = epi_plain(reinterpret_cast<float*>(counts), 0, G);     // c is never written in count mode
    E.apply_exp = apply_exp;
    E.cnt_mode = larger_is_better ? 1 : 2;
    E.cnt_off = pos_off; E.cnt_thr = thr; E.cnt_out = counts;
    {
        const int sr = score_split(Q, ld_q, nq, U, ld_u, G, r, E, sws, sws_bytes, u_packed, (hipStream_t)stream);
        if (sr != 0) return sr < 0 ? sr : TXE_OK;
    }
    return gemm_nt(A, B, E, nq, G, r, 1, (hipStream_t)stream);
}

// Fused scoring + best-k selection of one block of the loop (infer.py:96-106, test_fast.py:121-131: `sorted(enumerate(scores))[:k]`):
// the score tile never leaves the workgroup; every 128 x 128 tile leaves each row's k best columns at part_key / part_idx
// [nq][txe_score_topk_tiles(G)][k] (scratch), txe_topk_merge picks each query's best k among them: out_idx [nq][k] candidate rows
// (+ idx_base: the first row of a candidate shard), out_key [nq][k] their keys (score, negated when smaller is better; may be NULL).
// The same MFMA kernel and k order as txe_score_block: the scores behind the selection are bit-identical to the materialised ones.
// Ties in Python's stable order (ascending candidate row), NaN last.  1 <= k <= 8; G >= 1.
// floor_ws [nq] ints: scratch (the rows' rising selection floors, initialised here).
int txe_score_topk_tiles(int G) { return (G + 127) / 128; }
}  // extern "C"
namespace txe {
__global__ void topk_floor_init_kernel(int* __restrict__ f, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = topk_ord(-INFINITY);
}
}  // namespace txe
extern "C" {

int txe_score_topk_block(const float* Q, long long ld_q, int nq, const float* U, long long ld_u, int G, int r, int apply_exp,
                         int larger_is_better, int k, int idx_base, float* part_key, int* part_idx, int* floor_ws, int* out_idx,
                         float* out_key, void* sws, size_t sws_bytes, const void* u_packed, void* stream) {
    if (nq < 0 || G < 1 || r < 1 || ld_u < r || k < 1 || k > TOPK_MAX || !Q || !U || !part_key || !part_idx || !floor_ws || !out_idx)
        return TXE_ERR_ARG;
    if (nq == 0) return TXE_OK;
    hipLaunchKernelGGL(topk_floor_init_kernel, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, floor_ws, nq);
    TXE_CHECK_LAUNCH();
    VMat A = vmat_plain(Q, ld_q, nq, r);
    VMat B = vmat_plain(U, ld_u, G, r);
    Epi E = epi_plain(part_key, 0, G);                              // c is never written in top-k mode
    E.apply_exp = apply_exp;
    E.cnt_mode = larger_is_better ? 4 : 5;
    E.topk_k = k; E.topk_key = part_key; E.topk_idx = part_idx; E.topk_floor = floor_ws;
    E.force_bn128 = 1;                                              // (the scratch layout counts 128-wide column tiles)
    int rc = score_split(Q, ld_q, nq, U, ld_u, G, r, E, sws, sws_bytes, u_packed, (hipStream_t)stream);
    if (rc < 0) return rc;
    if (rc == 0) rc = gemm_nt(A, B, E, nq, G, r, 1, (hipStream_t)stream);
    else rc = TXE_OK;
    if (rc) return rc;
    return txe_topk_merge(part_key, part_idx, nq, (long long)txe_score_topk_tiles(G) * k, k, idx_base, out_idx, out_key, stream);
}

}  // extern "C"
