// Products with one SHORT dimension on the fp32 MFMA -- the bilinear matcher's work on the query RUNS of a training batch
// (model_zoo.py:301-328 with the collate of data_loaders.py:9-28: 128 distinct query rows behind 4,096 pairs):
//     V  [U][l]  = Q Wm^T          T  [U][Kp] = V Wf            (forward)
//     dV [U][l]  = dT Wf^T         dWf [l][Kp] = V^T dT         dWm [l][r] = dV^T Q        (backward)
// 0.1 - 0.3 GFLOP each with U = 128: on the big GEMM's 128-row tiles such a product is 16 - 64 workgroups with a long serial k-loop
// (41 + 29 us measured), and the dot-product kernels that replaced them in round 3 / 4 walked their reductions in 8 - 63 dependent
// round trips (8 - 33 us each, ~95 us of a 0.98-ms step).  Here ONE WAVE owns a 32 x 32 tile of C over a k-slice, the KS waves of a
// workgroup split the reduction, their partial tiles meet in LDS and are added in wave order (deterministic):
//   * operands go straight from global memory (L2: they are a few MB) into the MFMA's lane layout, no LDS staging --
//     v_mfma_f32_32x32x2_f32 sums over k in any order as long as A and B agree on it, so lane (row i, half q) of a 16-k step owns
//     k = k0 + 8 q + e, e = 0..7: a k-contiguous operand row is read as 32 contiguous bytes per lane (two 16-byte loads), a k-major
//     operand as eight 4-byte loads that are contiguous ACROSS the 32 lanes of a half (128-byte lines);
//   * two steps (32 k) are in flight per wave (128 VGPRs: two 8-wave workgroups per CU): the loads of step s + 2 are issued right behind the MFMAs of step s;
//   * the runs may be counted on the device (RunsRef.n_dev): M or K is then read there, row tiles walk with a grid stride.
// Exact fp32 (the MFMA is an fmaf chain); a slice's k order is ascending, slices are added in ascending order.
#pragma once
#include "txe_common.h"

namespace txe {

// The runs of equal consecutive query rows: either given (compact distinct rows Qu [U][r] + run offsets, U known on the host) or found on
// the device in the stacked matrix E2 [G][r] itself (run u's row is row off[u] of E2, the number of runs is a device scalar).
struct RunsRef {
    const int* off;       // [n + 1] first pair of every run, off[n] = G
    const int* n_dev;     // the number of runs on the device (NULL: n_host)
    int n_host;
    int first_row;        // 1: run u's distinct row = row off[u] of the stacked matrix;  0: row u of the compact matrix
};
__device__ __forceinline__ int runs_count(const RunsRef& R) { return R.n_dev ? *R.n_dev : R.n_host; }
__device__ __forceinline__ long long runs_row(const RunsRef& R, int u) { return R.first_row ? (long long)R.off[u] : (long long)u; }

typedef float sk_f32x16 __attribute__((ext_vector_type(16)));

struct SkinnyArgs {
    const float* A; long long lda;     // variant & 1 (A k-major): A[k][m], else A[m][k] (k contiguous)
    const float* B; long long ldb;     // variant & 2 (B k-major): B[k][n], else B[n][k]
    float* C; long long ldc;           // C [M][N]
    int M, N, K;
    int m_dyn, k_dyn;                  // M / K = the number of runs (read on the device when R.n_dev)
    int a_rows_first;                  // A's row m is row R.off[m] of its matrix (the stacked query matrix; k-contiguous A only)
    int b_k_first;                     // B's k-th row is row R.off[k] of its matrix (k-major B only)
    int variant;                       // bit 0: A k-major, bit 1: B k-major, bits 2-3: log2 of the vector width of the k-contiguous loads, bit 4: b_k_first
    int ks;                            // waves (k-slices) per tile; the launch's waves per workgroup are a multiple of it
    int nb;                            // workgroups of this job: ceil(nb_n * nb_m / (waves per workgroup / ks))
    int nb_n, nb_m;                    // column tiles, row tiles started by the grid (further row tiles: grid stride)
    RunsRef R;
};

#ifndef TXE_SK_NB
#define TXE_SK_NB 2
#endif
constexpr int SK_NB = TXE_SK_NB;               // 16-k steps in flight per wave

// eight operand values of lane (i, q) for the step at kb = k0 + 8 q: element e <-> k = kb + e.  `base` is the operand's (uniform)
// pointer, `lane` the lane's fixed 32-bit element offset into it (its row's start for a k-contiguous operand, its column for a k-major
// one): one scalar base + one 32-bit VGPR offset per load instead of a 64-bit address pair each (the first version kept 64 of those
// alive across its four steps in flight: 256 VGPRs and spills).  Addresses are always valid (clamped); the caller zeroes what lies past
// the slice.
// IND (k-major only): the operand's k-th row is row krow_first[k] of its matrix -- a TEMPLATE parameter: as a run-time `?:` it became a
// branch around a dependent load in front of every operand load, each with its own s_waitcnt vmcnt(0) (the T product: 37 us).
template <bool KM, int VEC, bool IND>
__device__ __forceinline__ void sk_load8(const float* __restrict__ base, unsigned lane, int ld, int kb, int K, const int* __restrict__ krow_first,
                                         float* v) {
    if (KM) {
        unsigned row[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = min(kb + e, K - 1);
            row[e] = IND ? (unsigned)krow_first[k] : (unsigned)k;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = base[row[e] * (unsigned)ld + lane];
    } else if (VEC == 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned k = (kb + 4 * h < K) ? (unsigned)(kb + 4 * h) : 0u;      // (K % 4 == 0: a vector is wholly inside or wholly outside)
            const float4 t = *reinterpret_cast<const float4*>(base + (lane + k));
            v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
        }
    } else if (VEC == 2) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const unsigned k = (kb + 2 * h < K) ? (unsigned)(kb + 2 * h) : 0u;
            const float2 t = *reinterpret_cast<const float2*>(base + (lane + k));
            v[2 * h] = t.x; v[2 * h + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = base[lane + (unsigned)min(kb + e, K - 1)];
    }
}

template <bool AKM, bool BKM, int VEC, bool BIND>
__device__ __forceinline__ void skinny_job(const SkinnyArgs& a, const int bid, float* __restrict__ red_all) {
    // the workgroup's waves form blockDim / (64 ks) GROUPS of ks waves: a group owns one 32 x 32 tile, its waves split the reduction
    const int KS = a.ks, groups = (int)(blockDim.x >> 6) / KS;
    const int wv = threadIdx.x >> 6, g = wv / KS, w = wv - g * KS, ln = threadIdx.x & 63, li = ln & 31, q = ln >> 5;
    float* red = red_all + (long long)g * KS * 16 * 64;
    const int nr = (a.m_dyn || a.k_dyn) ? runs_count(a.R) : 0;
    const int M = a.m_dyn ? nr : a.M, K = a.k_dyn ? nr : a.K, N = a.N;
    if (M <= 0 || K <= 0) return;                                   // (block-uniform)
    const int kc = (((K + KS - 1) / KS) + 15) & ~15;                // the wave's k-slice: whole 16-k steps
    const int k_lo = w * kc, k_hi = min(K, k_lo + kc);
    const int t = bid * groups + g;                                 // the group's tile: column tile t % nb_n, first row tile t / nb_n
    const bool idle = t >= a.nb_n * a.nb_m;
    const int tn = (t % a.nb_n) * 32;
    const int cj = min(tn + li, N - 1);
    const unsigned lb = BKM ? (unsigned)cj : (unsigned)cj * (unsigned)a.ldb;          // (operands below 2^31 elements: checked on the host)
    const int* bk_first = a.R.off;
    const int gt = threadIdx.x - g * KS * 64;                       // thread index inside the group
    for (int it = 0; it * 32 * a.nb_m < M; ++it) {                  // (the same trip count for every group: the barriers below are workgroup-wide)
        const int tm = ((t / a.nb_n) + it * a.nb_m) * 32;
        const bool work = !idle && tm < M;
        if (work) {
            const int ri = min(tm + li, M - 1);
            const unsigned la = AKM ? (unsigned)ri : (unsigned)(a.a_rows_first ? a.R.off[ri] : ri) * (unsigned)a.lda;
            sk_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            float av[SK_NB][8], bv[SK_NB][8];
#pragma unroll
            for (int s = 0; s < SK_NB; ++s) {
                sk_load8<AKM, VEC, false>(a.A, la, (int)a.lda, k_lo + 16 * s + 8 * q, K, nullptr, av[s]);
                sk_load8<BKM, VEC, BIND>(a.B, lb, (int)a.ldb, k_lo + 16 * s + 8 * q, K, bk_first, bv[s]);
            }
            __builtin_amdgcn_sched_barrier(0);                      // (left alone the scheduler sinks the loads to just before their use)
            for (int k0 = k_lo; k0 < k_hi; k0 += 16 * SK_NB) {
#pragma unroll
                for (int s = 0; s < SK_NB; ++s) {
                    const int kb = k0 + 16 * s + 8 * q;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool live = kb + e < k_hi;            // (past the slice: both factors 0 -- a clamped re-read may hold anything)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(live ? av[s][e] : 0.f, live ? bv[s][e] : 0.f, acc, 0, 0, 0);
                    }
                    sk_load8<AKM, VEC, false>(a.A, la, (int)a.lda, kb + 16 * SK_NB, K, nullptr, av[s]);
                    sk_load8<BKM, VEC, BIND>(a.B, lb, (int)a.ldb, kb + 16 * SK_NB, K, bk_first, bv[s]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + ln] = acc[r];
        }
        __syncthreads();
        if (work) {                                                 // the waves' partial tiles, added in wave (= k-slice) order
            for (int idx = gt; idx < 16 * 64; idx += KS * 64) {
                const int r = idx >> 6, l2 = idx & 63;
                float s = 0.f;
                for (int x = 0; x < KS; ++x) s += red[(x * 16 + r) * 64 + l2];
                const int row = tm + (r & 3) + 8 * (r >> 2) + 4 * (l2 >> 5), col = tn + (l2 & 31);
                if (row < M && col < N) a.C[(long long)row * a.ldc + col] = s;
            }
        }
        __syncthreads();                                            // (the next row tile overwrites the partials)
    }
}

__device__ __forceinline__ void skinny_dispatch(const SkinnyArgs& a, const int bid, float* red) {
    switch (a.variant) {                                            // (block-uniform)
        case 0: skinny_job<false, false, 1, false>(a, bid, red); break;    // NT: both operands k-contiguous, scalar loads
        case 4: skinny_job<false, false, 2, false>(a, bid, red); break;    //     8-byte loads
        case 8: skinny_job<false, false, 4, false>(a, bid, red); break;    //     16-byte loads
        case 2: skinny_job<false, true, 1, false>(a, bid, red); break;     // NN: B [k][n]
        case 6: skinny_job<false, true, 2, false>(a, bid, red); break;
        case 10: skinny_job<false, true, 4, false>(a, bid, red); break;
        case 3: skinny_job<true, true, 1, false>(a, bid, red); break;      // TN: A [k][m], B [k][n]
        default: skinny_job<true, true, 1, true>(a, bid, red); break;      // TN, B's k-th row through the runs (variant 19)
    }
}

constexpr int SK_MAXJOBS = 2;
struct SkinnyMulti { int n; SkinnyArgs j[SK_MAXJOBS]; };

// independent products in one launch, each on its own range of workgroups (all with the same number of waves)
__global__ __launch_bounds__(1024) void skinny_gemm_kernel(const SkinnyMulti m) {
    extern __shared__ float sk_red[];
    int b = blockIdx.x, i = 0;
    while (i + 1 < m.n && b >= m.j[i].nb) { b -= m.j[i].nb; ++i; }
    skinny_dispatch(m.j[i], b, sk_red);
}

// host side: geometry of one product.  m_hint / k_hint: the expected size of a run-counted dimension (the launch shape only; any count is
// handled).  The vector width of the k-contiguous loads follows from pointers / pitches / K.
static inline int skinny_pick_vec(const float* p, long long ld, int K) {
    if ((((uintptr_t)p) & 15) == 0 && (ld & 3) == 0 && (K & 3) == 0) return 4;
    if ((((uintptr_t)p) & 7) == 0 && (ld & 1) == 0 && (K & 1) == 0) return 2;
    return 1;
}
static inline void skinny_setup(SkinnyArgs& a, bool akm, bool bkm, int m_hint, int k_hint) {
    int vec = 1;
    if (!akm && !bkm) {
        const int va = skinny_pick_vec(a.A, a.lda, a.K), vb = skinny_pick_vec(a.B, a.ldb, a.K);
        vec = va < vb ? va : vb;
    } else if (!akm) {
        vec = skinny_pick_vec(a.A, a.lda, a.K);
    }
    if (a.k_dyn) vec = 1;                                           // (a run-counted K only occurs with both operands k-major)
    a.variant = (akm ? 1 : 0) | (bkm ? 2 : 0) | (vec == 4 ? 8 : (vec == 2 ? 4 : 0)) | ((akm && bkm && a.b_k_first) ? 16 : 0);
    a.nb_n = (a.N + 31) / 32;
    int nbm = (m_hint + 31) / 32;
    if (nbm < 1) nbm = 1;
    if (nbm > 64) nbm = 64;
    a.nb_m = nbm;
    // k-slices per tile: enough waves to put ~4 on every SIMD of the chip, at least two 16-k steps per wave
    const long long tiles = (long long)a.nb_n * a.nb_m;
    int ks = 1;
    while (ks < 16 && tiles * ks < 2048 && k_hint / (2 * ks) >= 32) ks *= 2;
    a.ks = ks;
}
// launch of up to SK_MAXJOBS set-up products: the workgroups hold max ks waves, a job with fewer slices puts several tiles in a workgroup
static inline int skinny_launch(SkinnyMulti& m, hipStream_t st) {
    for (int i = 0; i < m.n; ++i) {                                 // the kernel addresses its operands with 32-bit element offsets
        const SkinnyArgs& j = m.j[i];
        const long long rows = (long long)(j.M > j.K ? j.M : j.K) > j.N ? (long long)(j.M > j.K ? j.M : j.K) : (long long)j.N;
        const long long ld = j.lda > j.ldb ? j.lda : j.ldb;
        if ((rows + 1) * ld >= (1LL << 31)) return TXE_ERR_ARG;
    }
    int wg = 1;
    for (int i = 0; i < m.n; ++i) wg = m.j[i].ks > wg ? m.j[i].ks : wg;
    int total = 0;
    for (int i = 0; i < m.n; ++i) {
        const int groups = wg / m.j[i].ks;
        m.j[i].nb = (m.j[i].nb_n * m.j[i].nb_m + groups - 1) / groups;
        total += m.j[i].nb;
    }
    if (total == 0) return TXE_OK;
    hipLaunchKernelGGL(skinny_gemm_kernel, dim3(total), dim3(64 * wg), (size_t)wg * 16 * 64 * sizeof(float), st, m);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // namespace txe
