// Rank extraction for the all-candidate scoring loop, on device.
// Reference: test_fast.py:16-22 (rearrange) + model/metric.py:7-31:
//     rank(p) = 1 + #{ g not in positives(q) : score[q][g]  >  score[q][p] }      (similarity, info_nce)
//     rank(p) = 1 + #{ g not in positives(q) : score[q][g]  <  score[q][p] }      (distance)
// i.e. positives never count against each other and ties do not count (strict inequality).
// One workgroup per query streams the G scores of its row once (coalesced), compares against the query's
// few positives held in LDS, and reduces integer counts -- exact, no floating point accumulation.
#include "txe_common.h"

namespace txe {

constexpr int RANK_MAXP = 64;   // positives handled per pass

__global__ __launch_bounds__(256) void rank_kernel(const float* __restrict__ S, long long ld_s, int G,
                                                   const int* __restrict__ pos_off, const int* __restrict__ pos_idx,
                                                   int larger_is_better, int* __restrict__ ranks) {
    __shared__ float s_sp[RANK_MAXP];
    __shared__ int s_cnt[RANK_MAXP];
    const int q = blockIdx.x;
    const float* row = S + (long long)q * ld_s;
    const int pb = pos_off[q], pe = pos_off[q + 1];
    // Rows on a 16-byte pitch (every caller in this repo): the thresholds of four positives sit in registers and ONE sweep of
    // the row with 16-byte loads, four in flight per thread, counts against all of them -- the row is read once per four
    // positives instead of once per positive with dword loads.
    if ((ld_s & 3) == 0 && (reinterpret_cast<uintptr_t>(S) & 15) == 0) {
        const float4* row4 = reinterpret_cast<const float4*>(row);
        const int g4n = G >> 2;
        for (int c0 = pb; c0 < pe; c0 += 4) {
            const int np = min(4, pe - c0);
            float sp[4];
            int cnt[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) sp[k] = row[pos_idx[c0 + min(k, np - 1)]];
            for (int g0 = 0; g0 < g4n; g0 += 4 * 256) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = row4[min(g0 + u * 256 + (int)threadIdx.x, g4n - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = g0 + u * 256 + (int)threadIdx.x < g4n;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = larger_is_better ? ((v[u].x > sp[k]) + (v[u].y > sp[k]) + (v[u].z > sp[k]) + (v[u].w > sp[k]))
                                                       : ((v[u].x < sp[k]) + (v[u].y < sp[k]) + (v[u].z < sp[k]) + (v[u].w < sp[k]));
                        cnt[k] += ok ? c : 0;
                    }
                }
            }
            for (int g = (g4n << 2) + threadIdx.x; g < G; g += 256) {      // ragged end of the row
                const float x = row[g];
#pragma unroll
                for (int k = 0; k < 4; ++k) cnt[k] += larger_is_better ? (x > sp[k]) : (x < sp[k]);
            }
            for (int j = pb + threadIdx.x; j < pe; j += 256) {             // positives never count against each other
                const float x = row[pos_idx[j]];
#pragma unroll
                for (int k = 0; k < 4; ++k) cnt[k] -= larger_is_better ? (x > sp[k]) : (x < sp[k]);
            }
            if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int c = cnt[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
                if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt[k], c);       // integer: order independent
            }
            __syncthreads();
            if ((int)threadIdx.x < np) ranks[c0 + threadIdx.x] = s_cnt[threadIdx.x] + 1;
            __syncthreads();
        }
        return;
    }
    for (int c0 = pb; c0 < pe; c0 += RANK_MAXP) {
        const int np = min(RANK_MAXP, pe - c0);
        if (threadIdx.x < np) { s_sp[threadIdx.x] = row[pos_idx[c0 + threadIdx.x]]; s_cnt[threadIdx.x] = 0; }
        __syncthreads();
        for (int k = 0; k < np; ++k) {
            const float sp = s_sp[k];
            int cnt = 0;
            for (int g = threadIdx.x; g < G; g += blockDim.x) {
                const float v = row[g];
                cnt += larger_is_better ? (v > sp) : (v < sp);
            }
            // positives of this query are excluded from the comparison set
            for (int j = pb + threadIdx.x; j < pe; j += blockDim.x) {
                const float v = row[pos_idx[j]];
                cnt -= larger_is_better ? (v > sp) : (v < sp);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt[k], cnt);   // integer: order independent
        }
        __syncthreads();
        if (threadIdx.x < np) ranks[c0 + threadIdx.x] = s_cnt[threadIdx.x] + 1;
        __syncthreads();
    }
}

// ranks[j] = 1 + counts[j] - #{ j' of the same query : thr[j'] strictly better than thr[j] }   (positives never count against
// each other, metric.py:7-31)
__global__ void rank_finalize_kernel(const int* __restrict__ pos_off, int nq, const float* __restrict__ thr, const int* __restrict__ counts,
                                     int larger_is_better, int* __restrict__ ranks) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int pb = pos_off[q], pe = pos_off[q + 1];
    for (int j = pb; j < pe; ++j) {
        int c = counts[j];
        for (int k = pb; k < pe; ++k) c -= larger_is_better ? (thr[k] > thr[j]) : (thr[k] < thr[j]);
        ranks[j] = c + 1;
    }
}

// Best k entries of every row among `cnt` candidates (key, column) -- the per-tile lists of the score GEMM's top-k epilogue, or the
// per-rank lists of a candidate-sharded loop -- in Python's stable sorted() order: better key first, equal keys by ascending column.
// One wave per row: every lane folds its strided share into a best-k list (registers), then k rounds of a wave-wide arg-best over the
// lanes' heads; the lane that held the winner pops it.  Deterministic: no atomics, no dependence on the order of arrival.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ pkey, const int* __restrict__ pidx, int nq, long long cnt, int k,
                                                         int idx_base, int* __restrict__ out_idx, float* __restrict__ out_key) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (q >= nq) return;
    const float* kp = pkey + (long long)q * cnt;
    const int* ip = pidx + (long long)q * cnt;
    float bk[TOPK_MAX];
    int bi[TOPK_MAX];
    topk_init(bk, bi);
    float wk = -INFINITY;                                        // the lane's current k-th best
    int wi = 0x7fffffff;
    for (long long e0 = 0; e0 < cnt; e0 += 4 * 64) {            // four independent loads in flight per lane
        float kv[4];
        int iv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long e = e0 + u * 64 + l;
            const long long ec = e < cnt ? e : cnt - 1;
            kv[u] = kp[ec]; iv[u] = (e < cnt) ? ip[ec] : 0x7fffffff;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (iv[u] != 0x7fffffff && topk_better(kv[u], iv[u], wk, wi)) {
                topk_insert(bk, bi, kv[u], iv[u]);
                topk_kth(bk, bi, k, wk, wi);
            }
    }
    for (int t = 0; t < k; ++t) {
        float wk = bk[0];
        int wi = bi[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ok = __shfl_xor(wk, o, 64);
            const int oi = __shfl_xor(wi, o, 64);
            const bool take = topk_better(ok, oi, wk, wi);
            wk = take ? ok : wk; wi = take ? oi : wi;
        }
        if (bi[0] == wi && wi != 0x7fffffff) {                  // (columns are unique: exactly one lane holds the winner) pop it
#pragma unroll
            for (int j = 0; j + 1 < TOPK_MAX; ++j) { bk[j] = bk[j + 1]; bi[j] = bi[j + 1]; }
            bk[TOPK_MAX - 1] = -INFINITY; bi[TOPK_MAX - 1] = 0x7fffffff;
        }
        if (l == 0) {
            out_idx[(long long)q * k + t] = (wi == 0x7fffffff) ? -1 : wi + idx_base;
            if (out_key) out_key[(long long)q * k + t] = wk;
        }
    }
}

}  // namespace txe

using namespace txe;

extern "C" {

// keys / idx [nq][cnt] -> out_idx [nq][k] (+ idx_base; -1 where a row holds fewer than k real entries), out_key [nq][k] (may be NULL).
// Entries with idx == INT_MAX are empty slots.  1 <= k <= 8.
int txe_topk_merge(const float* keys, const int* idx, int nq, long long cnt, int k, int idx_base, int* out_idx, float* out_key, void* stream) {
    if (nq < 0 || cnt < 0 || k < 1 || k > TOPK_MAX || !keys || !idx || !out_idx) return TXE_ERR_ARG;
    if (nq == 0) return TXE_OK;
    if (cnt == 0) {
        if (hipMemsetAsync(out_idx, 0xff, (size_t)nq * k * sizeof(int), (hipStream_t)stream) != hipSuccess) return TXE_ERR_LAUNCH;
        return TXE_OK;
    }
    ProfScope prof("topk_merge_kernel", (hipStream_t)stream, 8.0 * nq * (double)cnt, 1);
    hipLaunchKernelGGL(topk_merge_kernel, dim3((nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, keys, idx, nq, cnt, k, idx_base, out_idx, out_key);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// S [nq][G] (row stride ld_s); pos_off [nq+1], pos_idx [pos_off[nq]] = candidate columns of each query's true
// parents (duplicates not allowed); ranks [pos_off[nq]] int32 out.
int txe_rank_block(const float* S, long long ld_s, int nq, int G, const int* pos_off, const int* pos_idx, int* ranks,
                   int larger_is_better, void* ws_unused, void* stream) {
    (void)ws_unused;
    if (nq < 0 || G < 0 || !S || !pos_off || !pos_idx || !ranks) return TXE_ERR_ARG;
    if (nq == 0) return TXE_OK;
    hipLaunchKernelGGL(rank_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, S, ld_s, G, pos_off, pos_idx, larger_is_better, ranks);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_rank_finalize(const int* pos_off, int nq, const float* thr, const int* counts, int larger_is_better, int* ranks, void* stream) {
    if (nq < 0 || !pos_off || !thr || !counts || !ranks) return TXE_ERR_ARG;
    if (nq == 0) return TXE_OK;
    hipLaunchKernelGGL(rank_finalize_kernel, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos_off, nq, thr, counts,
                       larger_is_better, ranks);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"
