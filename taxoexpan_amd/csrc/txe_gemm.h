// fp32 MFMA GEMM for gfx950 with "virtual matrix" operand loaders and fused epilogues.
//
// Every dense product on TaxoExpan's hot path (GAT/GCN feature projections model_zoo.py:37,83,
// their two backward products, the bilinear match model_zoo.py:313,328 and the all-candidate
// scoring loop test_fast.py:116-123) goes through this one kernel:
//
//     C[m][n] = sum_kk  A(m,kk) * B(kk,n)          m < M, n < N, kk in [k0,k1)
//
// * math: v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak on MI355X -- there is no TF32/xf32 on
//   gfx950).  Block tile 128x128x32, 4 waves (2x2), each wave 2x2 MFMA tiles of 32x32 -> 64 accumulator
//   VGPRs.  Operands are staged global -> registers -> LDS (double buffered, one barrier per k-tile).
// * operands are *virtual* row-major matrices (VMat): the concat of node features with the position
//   embedding row (model_zoo.py:215), the feature dropout (model_zoo.py:82), row/column extensions that
//   carry the folded attention projections, per-row scales ... are synthesised by the loader, so none of
//   those tensors is ever materialised in HBM.
// * either operand may be read "k-contiguous" (A[m][kk], kk fastest) or "row-contiguous"
//   (A(m,kk) = Mat[kk][m]); that covers NT / NN / TN products without transposing in HBM.
#pragma once
#include "txe_common.h"

namespace txe {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Logical row-major matrix [rows][cols] assembled from up to three arrays:
//   cols [0, cols_main)      : p  (rows < rows_main)  or p3 (rows >= rows_main, row r-rows_main)
//   cols [cols_main, cols)   : p2[er*ld2 + c-cols_main], er = pos ? pos[r] : r      (table / 2nd matrix)
// then optionally * rowscale[r] and * dropout factor(seed, r*drop_ld + c).
struct VMat {
    const float* p;
    long long ld;
    int rows, cols;
    int cols_main;
    const float* p2;
    long long ld2;
    const int* pos;
    int rows_main;
    const float* p3;
    long long ld3;
    const float* rowscale;
    float drop_p, drop_scale;
    unsigned long long seed;
    long long drop_ld;
};

static inline VMat vmat_plain(const float* p, long long ld, int rows, int cols) {
    VMat m;
    m.p = p; m.ld = ld; m.rows = rows; m.cols = cols; m.cols_main = cols;
    m.p2 = nullptr; m.ld2 = 0; m.pos = nullptr;
    m.rows_main = rows; m.p3 = nullptr; m.ld3 = 0;
    m.rowscale = nullptr; m.drop_p = 0.f; m.drop_scale = 1.f; m.seed = 0; m.drop_ld = cols;
    return m;
}

// Output side.  Logical C [rows][cols]:
//   n <  cols_main : c [m*ldc  + n]
//   n >= cols_main : c2[m*ldc2 + n - cols_main]
// value = acc (* rowscale[m]) (* dropout factor(seed, m*drop_ld + n)) (* leaky'(act_src[m][n]) for n<cols_main)
//         (exp() if apply_exp).  Split-K: block z writes at c + z*split_stride (no extras expected).
struct Epi {
    float* c;
    long long ldc;
    int cols_main;
    float* c2;
    long long ldc2;
    const float* act_src;
    long long ld_act;
    float act_slope;
    float drop_p, drop_scale;
    unsigned long long seed;
    long long drop_ld;
    int drop_col0;             // dropout index column = n + drop_col0
    const float* rowscale;
    int apply_exp;
    long long split_stride;
};

static inline Epi epi_plain(float* c, long long ldc, int cols) {
    Epi e;
    e.c = c; e.ldc = ldc; e.cols_main = cols; e.c2 = nullptr; e.ldc2 = 0;
    e.act_src = nullptr; e.ld_act = 0; e.act_slope = 1.f;
    e.drop_p = 0.f; e.drop_scale = 1.f; e.seed = 0; e.drop_ld = 0; e.drop_col0 = 0;
    e.rowscale = nullptr; e.apply_exp = 0; e.split_stride = 0;
    return e;
}

template <int V>
__device__ __forceinline__ void vmat_load(const VMat& M, int r, int c, float* v) {
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    if (r >= M.rows || c >= M.cols) return;
    const float* row = (r < M.rows_main) ? (M.p + (long long)r * M.ld) : (M.p3 + (long long)(r - M.rows_main) * M.ld3);
    if (c + V <= M.cols_main) {
        if constexpr (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(row + c);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else if constexpr (V == 2) {
            const float2 t = *reinterpret_cast<const float2*>(row + c);
            v[0] = t.x; v[1] = t.y;
        } else {
            v[0] = row[c];
        }
    } else {
        const long long er = M.pos ? (long long)M.pos[r] : (long long)r;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int cc = c + e;
            if (cc < M.cols_main) v[e] = row[cc];
            else if (cc < M.cols) v[e] = M.p2[er * M.ld2 + (cc - M.cols_main)];
        }
    }
    if (M.rowscale) {
        const float s = M.rowscale[r];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= s;
    }
    if (M.drop_p > 0.f) {
        const unsigned long long base = (unsigned long long)r * (unsigned long long)M.drop_ld + (unsigned long long)c;
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= drop_factor(M.seed, base + e, M.drop_p, M.drop_scale);
    }
}

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 32, GEMM_THREADS = 256;
constexpr int GEMM_KPAD = GEMM_BK + 4;  // k-contiguous LDS row stride (floats): conflict-free ds_read_b128

// KC = true : operand tile is [R=128 rows][BK] read along k   (LDS [row][BK+4])
// KC = false: operand tile is [BK][R=128]      read along rows (LDS [k][128])
template <bool KC, int V>
__device__ __forceinline__ void stage_load(const VMat& M, int row0, int k0, float* regs) {
    const int t = threadIdx.x;
    if constexpr (KC) {
        constexpr int VPR = GEMM_BK / V;             // vectors per tile row
        constexpr int RPP = GEMM_THREADS / VPR;      // rows per pass
        constexpr int PASSES = GEMM_BM / RPP;
        const int kq = t % VPR, r = t / VPR;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) vmat_load<V>(M, row0 + r + p * RPP, k0 + kq * V, regs + p * V);
    } else {
        constexpr int VPR = GEMM_BM / V;             // vectors per k-row
        constexpr int KPP = GEMM_THREADS / VPR;      // k-rows per pass
        constexpr int PASSES = GEMM_BK / KPP;
        const int mq = t % VPR, kr = t / VPR;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) vmat_load<V>(M, k0 + kr + p * KPP, row0 + mq * V, regs + p * V);
    }
}

template <bool KC, int V>
__device__ __forceinline__ void stage_store(float* lds, const float* regs) {
    const int t = threadIdx.x;
    if constexpr (KC) {
        constexpr int VPR = GEMM_BK / V, RPP = GEMM_THREADS / VPR, PASSES = GEMM_BM / RPP;
        const int kq = t % VPR, r = t / VPR;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            float* d = lds + (r + p * RPP) * GEMM_KPAD + kq * V;
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] = regs[p * V + e];
        }
    } else {
        constexpr int VPR = GEMM_BM / V, KPP = GEMM_THREADS / VPR, PASSES = GEMM_BK / KPP;
        const int mq = t % VPR, kr = t / VPR;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            float* d = lds + (kr + p * KPP) * GEMM_BM + mq * V;
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] = regs[p * V + e];
        }
    }
}

// MFMA operand fragment for the 32-row sub-tile starting at row r0, k-group kb (8 k values):
// lane l supplies row (l&31) and k = kb*8 + (l>>5)*4 + s for step s = 0..3.  A and B use the same
// (lane-half, step) -> k map, so the products line up whatever the storage mode.
template <bool KC>
__device__ __forceinline__ void frag_load(const float* lds, int r0, int kb, float* f) {
    const int l = threadIdx.x & 63;
    const int row = r0 + (l & 31), kk = kb * 8 + (l >> 5) * 4;
    if constexpr (KC) {
        const float4 t = *reinterpret_cast<const float4*>(lds + row * GEMM_KPAD + kk);
        f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = lds[(kk + s) * GEMM_BM + row];
    }
}

template <bool AK, bool BKC, int V>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(VMat A, VMat B, const Epi E, const int M, const int N,
                                                                const int K, const int ksplit) {
    constexpr int ASZ = AK ? GEMM_BM * GEMM_KPAD : GEMM_BK * GEMM_BM;
    constexpr int BSZ = BKC ? GEMM_BN * GEMM_KPAD : GEMM_BK * GEMM_BN;
    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];
    float* As = smem;
    float* Bs = smem + 2 * ASZ;

    const int nbn = (N + GEMM_BN - 1) / GEMM_BN;
    const int ntiles = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, ntiles);   // XCD-contiguous tile order: row panels stay in one L2
    const int tm = lb / nbn, tn = lb % nbn;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const int kbeg = blockIdx.y * ksplit;
    const int kend = min(K, kbeg + ksplit);

    // clip the reduction range into the operands' own bounds (split-K and K tails read zeros)
    if (AK) { A.cols = min(A.cols, kend); A.cols_main = min(A.cols_main, kend); }
    else    { A.rows = min(A.rows, kend); }
    if (BKC) { B.cols = min(B.cols, kend); B.cols_main = min(B.cols_main, kend); }
    else     { B.rows = min(B.rows, kend); }

    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wr = w >> 1, wc = w & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[16], rb[16];
    const int nk = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
    if (nk > 0) {
        stage_load<AK, V>(A, m0, kbeg, ra);
        stage_load<BKC, V>(B, n0, kbeg, rb);
        stage_store<AK, V>(As, ra);
        stage_store<BKC, V>(Bs, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
            stage_load<AK, V>(A, m0, kbeg + (kt + 1) * GEMM_BK, ra);
            stage_load<BKC, V>(B, n0, kbeg + (kt + 1) * GEMM_BK, rb);
        }
        const float* a_l = As + cur * ASZ;
        const float* b_l = Bs + cur * BSZ;
#pragma unroll
        for (int kb = 0; kb < GEMM_BK / 8; ++kb) {
            float fa[2][4], fb[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) frag_load<AK>(a_l, wr * 64 + i * 32, kb, fa[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) frag_load<BKC>(b_l, wc * 64 + j * 32, kb, fb[j]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (more) {
            stage_store<AK, V>(As + (cur ^ 1) * ASZ, ra);
            stage_store<BKC, V>(Bs + (cur ^ 1) * BSZ, rb);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* cbase = E.c + (long long)blockIdx.y * E.split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wc * 64 + j * 32 + (l & 31);
            if (n >= N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
                if (m >= M) continue;
                float v = acc[i][j][e];
                if (E.rowscale) v *= E.rowscale[m];
                if (E.drop_p > 0.f)
                    v *= drop_factor(E.seed, (unsigned long long)m * (unsigned long long)E.drop_ld + (unsigned long long)(n + E.drop_col0),
                                     E.drop_p, E.drop_scale);
                if (n < E.cols_main) {
                    if (E.act_src) v *= (E.act_src[(long long)m * E.ld_act + n] > 0.f) ? 1.f : E.act_slope;
                    if (E.apply_exp) v = __expf(v);
                    cbase[(long long)m * E.ldc + n] = v;
                } else {
                    E.c2[(long long)m * E.ldc2 + (n - E.cols_main)] = v;
                }
            }
        }
    }
}

static inline int gcd_vec(long long x) { return (x % 4 == 0) ? 4 : ((x % 2 == 0) ? 2 : 1); }
static inline int ptr_vec(const void* p) {
    const uintptr_t a = (uintptr_t)p;
    return (a % 16 == 0) ? 4 : ((a % 8 == 0) ? 2 : 1);
}
static inline int vmat_vec(const VMat& m) {
    int v = 4;
    auto upd = [&](int x) { if (x < v) v = x; };
    if (m.p) { upd(gcd_vec(m.ld)); upd(ptr_vec(m.p)); }
    if (m.p3) { upd(gcd_vec(m.ld3)); upd(ptr_vec(m.p3)); }
    return v;
}

// Launch C = A*B.  splits > 1 => split-K over gridDim.y, block z stores at E.c + z*E.split_stride.
template <bool AK, bool BKC>
static inline int gemm_launch_layout(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits,
                                     hipStream_t stream) {
    if (M <= 0 || N <= 0) return TXE_OK;
    int v = vmat_vec(A);
    const int vb = vmat_vec(B);
    if (vb < v) v = vb;
    const int nbm = (M + GEMM_BM - 1) / GEMM_BM, nbn = (N + GEMM_BN - 1) / GEMM_BN;
    if (splits < 1) splits = 1;
    int ksplit = (K + splits - 1) / splits;
    ksplit = ((ksplit + GEMM_BK - 1) / GEMM_BK) * GEMM_BK;
    if (ksplit == 0) ksplit = GEMM_BK;
    dim3 grid(nbm * nbn, splits), block(GEMM_THREADS);
    if (v == 4) hipLaunchKernelGGL((gemm_kernel<AK, BKC, 4>), grid, block, 0, stream, A, B, E, M, N, K, ksplit);
    else if (v == 2) hipLaunchKernelGGL((gemm_kernel<AK, BKC, 2>), grid, block, 0, stream, A, B, E, M, N, K, ksplit);
    else hipLaunchKernelGGL((gemm_kernel<AK, BKC, 1>), grid, block, 0, stream, A, B, E, M, N, K, ksplit);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// defined in txe_gemm.hip (one translation unit instantiates the kernels)
int gemm_nt(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s);  // A[m][k], B[n][k]
int gemm_nn(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s);  // A[m][k], B[k][n]
int gemm_tn(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s);  // A[k][m], B[k][n]

}  // namespace txe
