// Dense feature projections of the propagation layers, forward and backward, on the fp32 MFMA GEMM.
//
//   GATLayer (model_zoo.py:82-85):   h = feat_drop(cat(x, P[pos]));  ft = h W^T;  a1 = <ft, attn_l>;  a2 = <ft, attn_r>
//   GCNLayer (model_zoo.py:35-37):   h = dropout(cat(x, P[pos]));    hw = h W
//
// MI355X-first restructuring (identical math, different association):
//  * the concat with the position embedding and the dropout are synthesised by the GEMM's operand loader
//    (txe_gemm.h VMat) -- cat(x, P[pos]) and the dropped copy never exist in HBM;
//  * the attention projections are folded into the same GEMM: a1 = h (W^T attn_l) -> 2H extra output
//    columns computed from 2H folded weight rows wa = [attn_l; attn_r] (x) W.  That removes the two
//    N x H x D passes of model_zoo.py:84-85 in forward AND their two passes in backward: the gradients
//    d a1, d a2 ride as 2H extra columns of the incoming gradient through the dX and dW GEMMs and are
//    unfolded on the (tiny) weight side:  dW += attn (x) d wa,  d attn = <d wa, W>.
#include <string.h>

#include "txe_gemm.h"
#include "txe_gather.h"
#include "txe_colsum.h"
#include "txe_dxpos.h"
#include "txe_gemm_split.h"

namespace txe {

constexpr int MAX_VOCAB = 8;

// wa[h][k]   = sum_d attn_l[h*D+d] * W[(h*D+d)*ldw + k]
// wa[H+h][k] = sum_d attn_r[h*D+d] * W[(h*D+d)*ldw + k]            (k < Kt)
// One workgroup per (row r, 64-column chunk): 64 columns x 16 d-groups, LDS tree over the d-groups.
constexpr int FOLD_DG = 16;
__device__ __forceinline__ void fold_attn_job(const int bx, const int r /* 0 .. 2H-1 */, const float* __restrict__ W, long long ldw, int Kt,
                                              const float* __restrict__ attn_l, const float* __restrict__ attn_r, int H, int D,
                                              float* __restrict__ wa, long long ld_wa) {
    __shared__ float red[FOLD_DG][64];
    const int h = r % H;
    const float* attn = ((r < H) ? attn_l : attn_r) + (long long)h * D;
    const float* Wh = W + (long long)h * D * ldw;
    const int kl = threadIdx.x & 63, dg = threadIdx.x >> 6;
    const int k = bx * 64 + kl;
    const int kc = (k < Kt) ? k : 0;
    float acc = 0.f;
#pragma unroll 4
    for (int d = dg; d < D; d += FOLD_DG) acc = fmaf(attn[d], Wh[(long long)d * ldw + kc], acc);
    red[dg][kl] = acc;
    __syncthreads();
    if (dg == 0 && k < Kt) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < FOLD_DG; ++g) s += red[g][kl];
        wa[(long long)r * ld_wa + k] = s;
    }
}
__global__ __launch_bounds__(64 * FOLD_DG) void fold_attn_kernel(const float* __restrict__ W, long long ldw, int Kt,
                                                                 const float* __restrict__ attn_l, const float* __restrict__ attn_r,
                                                                 int H, int D, float* __restrict__ wa, long long ld_wa) {
    fold_attn_job(blockIdx.x, blockIdx.y, W, ldw, Kt, attn_l, attn_r, H, D, wa, ld_wa);
}

// dwa[r][k] = sum_s part[s][F + r][k]     r < 2H
__device__ __forceinline__ void ext_rows_job(const int r, const int k, const float* __restrict__ part, int S, long long split_stride, int F,
                                             int ldp, float* __restrict__ dwa) {
    if (k >= ldp) return;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += part[(long long)s * split_stride + (long long)(F + r) * ldp + k];
    dwa[(long long)r * ldp + k] = acc;
}
__global__ void reduce_ext_rows_kernel(const float* __restrict__ part, int S, long long split_stride, int F, int H2, int ldp,
                                       float* __restrict__ dwa) {
    ext_rows_job(blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, part, S, split_stride, F, ldp, dwa);
}

// One workgroup per weight row f = h*D + d:
//   dW[f][k]    = sum_s part[s][f][k] + attn_l[f] * dwa[h][k] + attn_r[f] * dwa[H+h][k]
//   d_attn_l[f] = sum_k dwa[h][k]   * W[f][k]
//   d_attn_r[f] = sum_k dwa[H+h][k] * W[f][k]
struct UnfoldArgs {
    const float* part; int S; long long split_stride; const float* dwa; long long ldp; const float* W; long long ldw;
    const float *attn_l, *attn_r; int H, D, Kt; float* dW; long long ld_dw; float *d_attn_l, *d_attn_r;
};
__device__ __forceinline__ void unfold_job(const int f, const UnfoldArgs& a) {
    __shared__ float red[2][4];
    const int h = f / a.D;
    const float al = a.attn_l[f], ar = a.attn_r[f];
    float dl = 0.f, dr = 0.f;
    for (int k = threadIdx.x; k < a.Kt; k += blockDim.x) {
        // the split-K partials are summed in slice order, eight (unconditional, clamped) loads in flight at a time
        const float* pp = a.part + (long long)f * a.ldp + k;
        const float gl = a.dwa[(long long)h * a.ldp + k], gr = a.dwa[(long long)(a.H + h) * a.ldp + k];
        const float wv = a.W[(long long)f * a.ldw + k];
        float acc = 0.f;
        for (int s0 = 0; s0 < a.S; s0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pp[(long long)min(s0 + j, a.S - 1) * a.split_stride];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (s0 + j < a.S) ? v[j] : 0.f;
        }
        a.dW[(long long)f * a.ld_dw + k] = acc + al * gl + ar * gr;
        dl = fmaf(gl, wv, dl);
        dr = fmaf(gr, wv, dr);
    }
    dl = wave_sum(dl);
    dr = wave_sum(dr);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = dl; red[1][w] = dr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.d_attn_l[f] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        a.d_attn_r[f] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// out[i] = sum_s part[s*stride + i]
__global__ void reduce_splits_kernel(const float* __restrict__ part, int S, long long stride, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += part[(long long)s * stride + i];
        out[i] = acc;
    }
}

// Deterministic two-stage "sum rows by position class":  dP[c][j] = sum_{m : pos[m]==c} x[m][j]
// stage 1: block b owns rows [b*rows_per_block, ...): 64 column lanes x 4 row groups, fixed-order LDS combine.
struct Seg1Args { const float* x; long long ldx; int cols; float* part; };
__device__ __forceinline__ void segsum1_job(const int bid, const Seg1Args& a, const int* __restrict__ pos, int n_rows, int vocab,
                                            int rows_per_block) {
    __shared__ float red[4][MAX_VOCAB][64];
    const int r0 = bid * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
    const int jl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    for (int j0 = 0; j0 < a.cols; j0 += 64) {
        const int j = j0 + jl;
        const int jc = (j < a.cols) ? j : 0;
        float acc[MAX_VOCAB];
#pragma unroll
        for (int c = 0; c < MAX_VOCAB; ++c) acc[c] = 0.f;
#pragma unroll 4
        for (int m = r0 + rg; m < r1; m += 4) {
            const int pc = pos[m];
            const float v = a.x[(long long)m * a.ldx + jc];
#pragma unroll
            for (int c = 0; c < MAX_VOCAB; ++c) acc[c] += (pc == c) ? v : 0.f;
        }
#pragma unroll
        for (int c = 0; c < MAX_VOCAB; ++c) red[rg][c][jl] = acc[c];
        __syncthreads();
        if (rg == 0 && j < a.cols)
            for (int c = 0; c < vocab; ++c)
                a.part[((long long)bid * vocab + c) * a.cols + j] = red[0][c][jl] + red[1][c][jl] + red[2][c][jl] + red[3][c][jl];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void pos_segsum_stage1(const float* __restrict__ x, long long ldx, const int* __restrict__ pos,
                                                         int n_rows, int cols, int vocab, int rows_per_block,
                                                         float* __restrict__ part /*[nb][vocab][cols]*/) {
    Seg1Args a{x, ldx, cols, part};
    segsum1_job(blockIdx.x, a, pos, n_rows, vocab, rows_per_block);
}
struct Seg2Args { const float* part; int nb; int n /* vocab * cols */; float* out; };
__device__ __forceinline__ void segsum2_job(const int bid, const Seg2Args& a) {
    __shared__ float red[4][64];
    const int il = threadIdx.x & 63, bg = threadIdx.x >> 6;
    const int i = bid * 64 + il;
    const int ic = (i < a.n) ? i : 0;
    float acc = 0.f;
    for (int b0 = bg; b0 < a.nb; b0 += 64) {          // 16 partial rows of this row group per step: clamped loads issued together
        float v[16];                                   // (the fused sweeps leave ~1,100 partial rows: one load in flight per thread
#pragma unroll                                         //  made this walk 280 dependent round trips)
        for (int q = 0; q < 16; ++q) v[q] = a.part[(long long)min(b0 + 4 * q, a.nb - 1) * a.n + ic];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += (b0 + 4 * q < a.nb) ? v[q] : 0.f;
    }
    red[bg][il] = acc;
    __syncthreads();
    if (bg == 0 && i < a.n) a.out[i] = red[0][il] + red[1][il] + red[2][il] + red[3][il];
}
__global__ __launch_bounds__(256) void pos_segsum_stage2(const float* __restrict__ part, int nb, int vocab, int cols,
                                                         float* __restrict__ out) {
    Seg2Args a{part, nb, vocab * cols, out};
    segsum2_job(blockIdx.x, a);
}

// The reductions that end a GATLayer's backward, as TWO launches of independent jobs on disjoint workgroup ranges:
//   phase A: per-block partial position sums (embedding gradient; readout position-weight gradient) and the folded attention rows'
//            gradient d_wa (split-K slices of the extension rows, or the per-block partials of the folded output layer);
//   phase B: dW / d_attn from d_wa (unfold) and the second stage of the position sums.
struct TailA {
    int nb_dx; DxPosArgs dx;                    // leading jobs: the streaming d_X kernel's row blocks (dxpos_finish_job)
    int nb_s1a, nb_s1b, nb_r, r_kind;           // r_kind 1: extension rows of the split-K weight gradient, 2: stage 2 over dwa_part
    Seg1Args s1a, s1b;
    const int* pos; int n_rows, vocab, rows_per_block;
    const float* rpart; int S; long long split_stride; int F, ldp, nbx; float* dwa;
    Seg2Args r2;
};
__device__ __forceinline__ void reduce_a_job(int b, const TailA& a) {
    if (b < a.nb_dx) { dxpos_finish_job(b, a.dx); return; }
    b -= a.nb_dx;
    if (b < a.nb_s1a) { segsum1_job(b, a.s1a, a.pos, a.n_rows, a.vocab, a.rows_per_block); return; }
    b -= a.nb_s1a;
    if (b < a.nb_s1b) { segsum1_job(b, a.s1b, a.pos, a.n_rows, a.vocab, a.rows_per_block); return; }
    b -= a.nb_s1b;
    if (a.r_kind == 1) ext_rows_job(b / a.nbx, (b % a.nbx) * 256 + threadIdx.x, a.rpart, a.S, a.split_stride, a.F, a.ldp, a.dwa);
    else segsum2_job(b, a.r2);
}
__global__ __launch_bounds__(256) void gat_bwd_reduce_a_kernel(const TailA a) { reduce_a_job(blockIdx.x, a); }
struct TailB {
    int nb_u, nb_2a, nb_2b;
    UnfoldArgs u;
    Seg2Args s2a, s2b;
};
__device__ __forceinline__ void tail_b_job(int b, const TailB& a) {
    if (b < a.nb_u) { unfold_job(b, a.u); return; }
    b -= a.nb_u;
    if (b < a.nb_2a) { segsum2_job(b, a.s2a); return; }
    segsum2_job(b - a.nb_2a, a.s2b);
}
__global__ __launch_bounds__(256) void gat_bwd_reduce_b_kernel(const TailB a) { tail_b_job(blockIdx.x, a); }

// Phase B of SEVERAL layers in one launch.  A layer's phase B only finishes parameter gradients (nothing downstream in the backward
// pass reads them), so a caller may DEFER it (phases | 64) into a host-side chain and let the last layer's call launch them all:
// one ~17 us dispatch per stack instead of one per layer.
constexpr int TAIL_CHAIN_MAX = 3;                       // deferred layers a chain holds (a fourth deferral flushes)
struct TailChain { int n; int pad; TailB tb[TAIL_CHAIN_MAX]; };
struct TailMulti { int n; int nb_end[TAIL_CHAIN_MAX + 1]; TailB tb[TAIL_CHAIN_MAX + 1]; };
__global__ __launch_bounds__(256) void gat_bwd_reduce_b_multi_kernel(const TailMulti m) {
    int b = blockIdx.x, i = 0;
    while (i + 1 < m.n && b >= m.nb_end[i]) ++i;                    // (block-uniform)
    tail_b_job(b - ((i > 0) ? m.nb_end[i - 1] : 0), m.tb[i]);
}
static inline int tail_b_blocks(const TailB& t) { return t.nb_u + t.nb_2a + t.nb_2b; }
// launch `own` (if given) together with everything the chain holds, or -- defer -- append `own` to the chain
static int tail_b_submit(const TailB* own, void* chain_, bool defer, hipStream_t s) {
    TailChain* c = reinterpret_cast<TailChain*>(chain_);
    if (c && (c->n < 0 || c->n > TAIL_CHAIN_MAX)) return TXE_ERR_ARG;
    if (defer && c && own && c->n < TAIL_CHAIN_MAX) { c->tb[c->n++] = *own; return TXE_OK; }
    TailMulti m;
    memset(&m, 0, sizeof(m));
    int total = 0;
    auto add = [&](const TailB& t) { if (tail_b_blocks(t) > 0) { m.tb[m.n] = t; total += tail_b_blocks(t); m.nb_end[m.n++] = total; } };
    if (own) add(*own);
    if (c) { for (int i = 0; i < c->n; ++i) add(c->tb[i]); c->n = 0; }
    if (m.n == 0) return TXE_OK;
    if (m.n == 1) hipLaunchKernelGGL(gat_bwd_reduce_b_kernel, dim3(total), dim3(256), 0, s, m.tb[0]);
    else hipLaunchKernelGGL(gat_bwd_reduce_b_multi_kernel, dim3(total), dim3(256), 0, s, m);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__device__ __forceinline__ void dropout_mask_job(const int bid, const int nb, long long n_words, unsigned long long seed, unsigned thr16,
                                                 unsigned* __restrict__ mask) {
    for (long long w = (long long)bid * blockDim.x + threadIdx.x; w < n_words; w += (long long)nb * blockDim.x)
        mask[w] = drop_mask_word(seed, (unsigned long long)w, thr16);
}
__global__ void dropout_mask_kernel(long long n_words, unsigned long long seed, unsigned thr16, unsigned* __restrict__ mask) {
    dropout_mask_job(blockIdx.x, gridDim.x, n_words, seed, thr16, mask);
}

}  // namespace txe

using namespace txe;

extern "C" {

// Feature-dropout keep mask for an [n_rows][n_cols] operand: bit (r, c) = word[r*ceil(n_cols/32) + c/32] >> (c%32) & 1.
// nn.Dropout(p) of model_zoo.py:36,82 -- generated once per layer per step, reused by forward, dX and dW.
size_t txe_dropout_mask_bytes(long long n_rows, int n_cols) { return (size_t)n_rows * ((n_cols + 31) / 32) * 4; }

int txe_dropout_mask(long long n_rows, int n_cols, float p, unsigned long long seed, unsigned* mask, void* stream) {
    if (n_rows < 0 || n_cols < 1 || p < 0.f || p >= 1.f || !mask) return TXE_ERR_ARG;
    const long long n_words = n_rows * ((n_cols + 31) / 32);
    if (n_words == 0) return TXE_OK;
    const unsigned thr16 = (unsigned)(p * 65536.0f + 0.5f);
    const int nb = (int)((n_words + 255) / 256 < 4096 ? (n_words + 255) / 256 : 4096);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n_words, seed, thr16, mask);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"

namespace txe {

// ---------------------------------------------------------------------------------------------
// GATLayer dense part on PADDED operands.
//   X  [N][Kp]   layer input:  [h (Kh) | Emb[pos] (Pd) | 0 ...],  Kp = roundup(Kh+Pd, 32).  The producer of h (the previous
//                layer's aggregation kernel, or txe_gat_build_x for the raw features) writes straight into it.
//   Wp [Fp][Kp]  packed weights: rows < F = fc.weight, rows F..F+2H = folded attention rows, rest 0; Fp = roundup(F+2H,128)
//   Y  [N][Fp]   projection output: [ft (F) | a1 (H) | a2 (H) | unused];  d_Y has the same layout with ZERO padding.
// Every GEMM operand is then a plain, 16-byte aligned, tile-padded matrix (all tiles take the hoisted fast path); the only
// loader-side extra left is the dropout bit mask on X.
// ---------------------------------------------------------------------------------------------
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Wp[f][k] = W[f][k] (f < F, k < Kt) else 0   (rows F..Fe are written by fold_attn_kernel)
// One thread per 4 consecutive packed columns (Kp % 4 == 0), two such quads in flight per thread: every load is issued before the
// first store, the packed row leaves as 16-byte stores.  (One wave per row with a scalar column loop made the 2,080-column rows of
// the output layer a 33-deep load -> store chain per wave.)
constexpr int PREP_U = 2;
// extra (or NULL): row F of Wp = extra[0 .. Kt) -- a folded GCN layer's bias rides as one more weight row (txe_gcn_layer_prepare)
__device__ __forceinline__ void pack_w_job(const int bid, const int nb, const float* __restrict__ W, int F, int Fe, int Fp, int Kt, int Kp,
                                           float* __restrict__ Wp, const float* __restrict__ extra = nullptr) {
    const unsigned qpr = (unsigned)Kp >> 2, total = (unsigned)Fp * qpr;          // (a weight matrix: far below 2^32 quads)
    const bool v2 = ((Kt & 1) == 0) && (((uintptr_t)W & 7) == 0);                // rows 8-byte aligned: float2 loads
    for (unsigned base = (unsigned)bid * blockDim.x * PREP_U; base < total; base += (unsigned)nb * blockDim.x * PREP_U) {
        float v[PREP_U][4];
        unsigned fi[PREP_U], ki[PREP_U];
#pragma unroll
        for (int u = 0; u < PREP_U; ++u) {
            const unsigned i = base + u * blockDim.x + threadIdx.x;
            const unsigned f = i / qpr, k = (i - f * qpr) * 4;
            fi[u] = f; ki[u] = k;
            const bool isx = extra != nullptr && i < total && f == (unsigned)F;
            const bool live = (i < total && f < (unsigned)F) || isx;
            const float* src = isx ? extra : W + (long long)(live ? f : 0) * Kt;
            if (live && k + 3 < (unsigned)Kt && v2) {
                const float2 a = *reinterpret_cast<const float2*>(src + k), b = *reinterpret_cast<const float2*>(src + k + 2);
                v[u][0] = a.x; v[u][1] = a.y; v[u][2] = b.x; v[u][3] = b.y;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = (live && k + e < (unsigned)Kt) ? src[k + e] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < PREP_U; ++u) {
            const unsigned i = base + u * blockDim.x + threadIdx.x;
            if (i >= total) continue;
            float* dst = Wp + (long long)fi[u] * Kp + ki[u];
            if (fi[u] >= (unsigned)F && fi[u] < (unsigned)Fe) {                   // folded rows: the other job writes [0, Kt); zero the padding
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ki[u] + e >= (unsigned)Kt) dst[e] = 0.f;
            } else {
                *reinterpret_cast<float4*>(dst) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
            }
        }
    }
}
__global__ void pack_w_kernel(const float* __restrict__ W, int F, int Fe, int Fp, int Kt, int Kp, float* __restrict__ Wp) {
    pack_w_job(blockIdx.x, gridDim.x, W, F, Fe, Fp, Kt, Kp, Wp);
}

// X[r][c] = h[r][c] (c < Kh, only when h != NULL) | P[pos[r]][c-Kh] (Kh <= c < Kt) | 0 (Kt <= c < Kp)
// drop_thr16 != 0: the feature dropout is applied HERE (X[r][c] *= keep(r, c) ? drop_scale : 0, the very bits of drop_mask_word over
// ceil(Kt/32) words per row): the layer's GEMMs then read X as a plain operand -- no mask words, no selects in their loaders (the
// first-layer projection and its weight gradient: 226 -> 213 us, 282 -> 269 us).  With `mask` given (only when h != NULL: every
// word of a row is then needed here anyway) the job also WRITES the keep mask, and the launch needs no mask job for this layer.
//
// A workgroup owns a block of consecutive rows; one thread per 4 consecutive columns (a "quad", Kp % 32 == 0), PREP_UX quads in
// flight per thread, 16-byte stores.  Branch-free: every quad issues its pos[] load and its (column-clamped) feature loads, the
// workgroup then hashes the rows' mask words ONCE each into LDS (under those loads' latency), and -- pos[] being the oldest load in
// flight -- the (clamped) position-embedding loads follow; selects afterwards: two memory latencies per thread whatever mix of
// feature / embedding / padding quads a wave holds.  (One wave per row with a scalar column loop ran 5 dependent load -> hash ->
// store rounds per wave; divergent quad kinds with the dependent pos -> P chain inside a branch were slower still.)
constexpr int PREP_UX = 4;
constexpr int PREP_LDSW = 4096;                         // mask words a row block may hold in LDS
__device__ __forceinline__ void build_x_job(const int bid, const int nb, const float* __restrict__ h, long long ld_h, const int* __restrict__ pos,
                                            const float* __restrict__ P, int n_rows, int Kh, int Pd, int Kp, float* __restrict__ X,
                                            const unsigned long long drop_seed = 0, const unsigned drop_thr16 = 0, const float drop_scale = 1.f,
                                            unsigned* __restrict__ mask = nullptr) {
    __shared__ unsigned s_words[PREP_LDSW];
    const int T = blockDim.x;
    const int Kt = Kh + Pd, wpr = (Kt + 31) >> 5;
    const int q0 = h ? 0 : (Kh >> 2);                   // h == NULL: columns [0, Kh) are in place already
    const int qn = (Kp >> 2) - q0;                      // quads per row handled here
    const int w_lo = (q0 * 4) >> 5, nw = (Kp >> 5) - w_lo;   // mask words of a row that cover those quads (Kp/32 >= wpr; words >= wpr: no bits)
    const bool drop = drop_thr16 != 0u;
    const bool lds_words = drop && nw <= PREP_LDSW;
    int RB = (T * PREP_UX) / qn;                        // rows per block: one pass of PREP_UX quads per thread
    if (RB < 1) RB = 1;
    if (lds_words && RB * nw > PREP_LDSW) RB = PREP_LDSW / nw;
    const bool v2 = h && Kh >= 2 && ((Kh & 1) == 0) && ((ld_h & 1) == 0) && (((uintptr_t)h & 7) == 0);   // whole 8-byte pairs inside a row
    for (long long r0l = (long long)bid * RB; r0l < n_rows; r0l += (long long)nb * RB) {
        const int r0 = (int)r0l, nr = min(RB, n_rows - r0), nq = nr * qn;
        for (int pb = 0; pb < nq; pb += T * PREP_UX) {  // (one pass unless a single row has more than T * PREP_UX quads)
            float hv[PREP_UX][4], pv[PREP_UX][4];
            int li[PREP_UX], ci[PREP_UX], pr[PREP_UX];  // local row (-1: no quad), first column, pos[]
#pragma unroll
            for (int u = 0; u < PREP_UX; ++u) {
                const int i = pb + u * T + (int)threadIdx.x;
                const bool live = i < nq;
                const int lr = live ? i / qn : 0;
                li[u] = live ? lr : -1;
                ci[u] = (q0 + (live ? i - lr * qn : 0)) * 4;
                pr[u] = (Pd > 0) ? pos[r0 + lr] : 0;
            }
            if (h) {
#pragma unroll
                for (int u = 0; u < PREP_UX; ++u) {
                    const float* hrow = h + (long long)(r0 + max(li[u], 0)) * ld_h;
                    const int c = ci[u];
                    if (v2) {
                        const float2 a = *reinterpret_cast<const float2*>(hrow + min(c, Kh - 2));
                        const float2 b = *reinterpret_cast<const float2*>(hrow + min(c + 2, Kh - 2));
                        hv[u][0] = a.x; hv[u][1] = a.y; hv[u][2] = b.x; hv[u][3] = b.y;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[u][e] = hrow[min(c + e, Kh - 1)];
                    }
                }
            }
            if (lds_words && pb == 0) {                 // the block's mask words, once each (under the loads above)
                for (int wi = threadIdx.x; wi < nr * nw; wi += T) {
                    const int lr = wi / nw, wl = w_lo + (wi - lr * nw);
                    const unsigned long long w = (unsigned long long)(r0 + lr) * wpr + wl;
                    const unsigned word = (wl < wpr) ? drop_mask_word(drop_seed, w, drop_thr16) : 0u;
                    s_words[wi] = word;
                    if (mask && wl < wpr) mask[w] = word;
                }
                __syncthreads();
            }
            if (Pd > 0) {
#pragma unroll
                for (int u = 0; u < PREP_UX; ++u) {
                    const float* prow = P + (long long)pr[u] * Pd;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv[u][e] = prow[min(max(ci[u] + e - Kh, 0), Pd - 1)];
                }
            }
#pragma unroll
            for (int u = 0; u < PREP_UX; ++u) {
                if (li[u] < 0) continue;
                const int r = r0 + li[u], c = ci[u];
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (c + e < Kh) ? (h ? hv[u][e] : 0.f) : ((c + e < Kt) ? pv[u][e] : 0.f);
                if (drop && c < Kt) {                   // bits (c & 31) .. +3 of the row's mask word c / 32
                    const unsigned word = lds_words ? s_words[li[u] * nw + (c >> 5) - w_lo]
                                                    : drop_mask_word(drop_seed, (unsigned long long)r * wpr + (c >> 5), drop_thr16);
#pragma unroll
                    for (int e = 0; e < 4; ++e)         // (columns in [Kt, Kp) hold zeros: scaling them is harmless)
                        v[e] = ((word >> ((c & 31) + e)) & 1u) ? v[e] * drop_scale : 0.f;
                }
                float* dst = X + (long long)r * Kp + c;
                if (!h && c < Kh) {                     // the quad straddling Kh: its feature columns belong to the layer below
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e >= Kh) dst[e] = v[e];
                } else {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        if (lds_words) __syncthreads();                 // (the next row block overwrites the words)
    }
}
__global__ void build_x_kernel(const float* __restrict__ h, long long ld_h, const int* __restrict__ pos, const float* __restrict__ P,
                               int n_rows, int Kh, int Pd, int Kp, float* __restrict__ X) {
    build_x_job(blockIdx.x, gridDim.x, h, ld_h, pos, P, n_rows, Kh, Pd, Kp, X);
}

// Everything a GATLayer needs before its projection GEMM, in ONE launch (four independent jobs on disjoint ranges of workgroups):
// layer input X (build_x), packed weights Wp (pack_w), folded attention rows (fold_attn), feature-dropout keep mask.
// workgroups of build_x_job: one per block of rows (the device code's RB)
static inline int build_x_blocks(int T, int n_rows, int Kh, int Pd, int Kp, bool has_h, bool drop, int cap) {
    const int q0 = has_h ? 0 : (Kh >> 2), qn = (Kp >> 2) - q0, nw = (Kp >> 5) - ((q0 * 4) >> 5);
    if (qn <= 0 || n_rows <= 0) return 0;
    int RB = (T * PREP_UX) / qn;
    if (RB < 1) RB = 1;
    if (drop && nw <= PREP_LDSW && RB * nw > PREP_LDSW) RB = PREP_LDSW / nw;
    const long long b = ((long long)n_rows + RB - 1) / RB;
    return (int)(b < cap ? b : cap);
}
struct PrepArgs {
    int nb_x, nb_w, nb_f, nb_m, fold_bx;
    const float* h; long long ld_h; const int* pos; const float* P; int n_rows, Kh, Pd, Kp; float* X;
    const float *W, *attn_l, *attn_r; int H, D, F, Fe, Fp, Kt; float* Wp;
    int pk_rows, pk_ext, pk_prows, pk_cols, pk_pcols;       // packing job: W [pk_rows][pk_cols] -> Wp [pk_prows][pk_pcols], rows [pk_rows, pk_ext) left to fold
    const float* pk_extra;                                  // ... or NULL / one more row (row pk_rows) to pack behind W
    long long n_words; unsigned long long seed; unsigned thr16; unsigned* mask;
    int x_dropped; float drop_scale;                        // build_x applies the dropout itself (the mask is still written:
    int x_mask;                                             //  by build_x too when x_mask, else by the mask job)
};
__device__ __forceinline__ void prepare_jobs(const PrepArgs& a, int b) {
    // the latency-bound job (a strided reduction per folded row) is dispatched first, the streaming jobs fill in behind it
    if (b < a.nb_f) {
        fold_attn_job(b % a.fold_bx, b / a.fold_bx, a.W, (long long)a.Kt, a.Kt, a.attn_l, a.attn_r, a.H, a.D, a.Wp + (long long)a.F * a.Kp,
                      (long long)a.Kp);
        return;
    }
    b -= a.nb_f;
    if (b < a.nb_w) { pack_w_job(b, a.nb_w, a.W, a.pk_rows, a.pk_ext, a.pk_prows, a.pk_cols, a.pk_pcols, a.Wp, a.pk_extra); return; }
    b -= a.nb_w;
    if (b < a.nb_m) { dropout_mask_job(b, a.nb_m, a.n_words, a.seed, a.thr16, a.mask); return; }
    b -= a.nb_m;
    build_x_job(b, a.nb_x, a.h, a.ld_h, a.pos, a.P, a.n_rows, a.Kh, a.Pd, a.Kp, a.X, a.seed, a.x_dropped ? a.thr16 : 0u, a.drop_scale,
                a.x_mask ? a.mask : nullptr);
}
__global__ __launch_bounds__(64 * FOLD_DG) void gat_prepare_kernel(const PrepArgs a) { prepare_jobs(a, blockIdx.x); }

// zero columns [c0, c1) of a row-major [n_rows][ld] matrix
__global__ void zero_cols_kernel(float* __restrict__ x, long long ld, int n_rows, int c0, int c1) {
    const int wdt = c1 - c0;
    const long long n = (long long)n_rows * wdt;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        x[(i / wdt) * ld + c0 + (i % wdt)] = 0.f;
}

// Y[v][c] = T[row[v]][c] + T2[row2[v]][c]   (16-byte columns): the layer-0 projection of a batch whose node features are rows of a
// taxonomy table -- T = table W^T computed once per DISTINCT taxonomy node, T2 = the 3 position-embedding rows' projections.
__global__ __launch_bounds__(256) void gather_add_rows_kernel(const float* __restrict__ T, long long ld_t, const int* __restrict__ row,
                                                              const float* __restrict__ T2, long long ld_t2, const int* __restrict__ row2,
                                                              long long n_rows, int nvec, float* __restrict__ Y, long long ld_y) {
    const long long total = n_rows * nvec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long v = i / nvec;
        const int j = (int)(i % nvec);
        float4 a = *reinterpret_cast<const float4*>(T + (long long)row[v] * ld_t + 4 * j);
        if (T2) {
            const float4 b = *reinterpret_cast<const float4*>(T2 + (long long)row2[v] * ld_t2 + 4 * j);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4*>(Y + v * ld_y + 4 * j) = a;
    }
}

struct DenseWs {
    float* dwa;     // [2H][Kp]
    float* dxpart;  // the streaming d_X kernel's k-slice partial products
    float* ppart;   // [nb][vocab][Pd]
    float* part;    // [S][Fp][Kp]
    void* tail;
    size_t tail_bytes;
    int splits, seg_blocks, seg_rows;
    size_t total;
};

static DenseWs plan_dense_ws(void* ws, int n, int Fp, int H2, int Kp, int Pd, int vocab, bool dx_stream = true, int min_splits = 0) {
    DenseWs p;
    char* b = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) { float* r = (float*)(b + off); off += align_up(bytes, 256); return r; };
    p.dwa = take((size_t)H2 * Kp * 4);
    p.seg_rows = 64;
    p.seg_blocks = (n + p.seg_rows - 1) / p.seg_rows;
    if (p.seg_blocks < 1) p.seg_blocks = 1;
    // (sized for the streaming d_X kernel's 16-row workgroups, which write these partial sums themselves)
    const int ppart_blocks = dxpos_blocks(n) > p.seg_blocks ? dxpos_blocks(n) : p.seg_blocks;
    p.ppart = take((size_t)ppart_blocks * (vocab > 0 ? vocab : 1) * (Pd > 0 ? Pd : 1) * 4);
    p.dxpart = take((Pd > 0 && dx_stream) ? dxpos_part_bytes(n, Fp) : 0);
    p.splits = choose_splits(Fp, Kp, n);
    p.part = take((size_t)(p.splits > min_splits ? p.splits : min_splits) * Fp * Kp * 4);
    p.tail_bytes = gemm_tail_ws_bytes();
    p.tail = take(p.tail_bytes);
    p.total = off;
    return p;
}

// A GCNLayer's weight gradient dW [Kp][Fop] = X^T d_hw has its SHORT side first (Kp = 320 rows of 128-row tiles: 3 row panels, one of them
// half empty, x 64-wide column tiles: 96 us for 5.7 GFLOP on the training batch, 0.38 of the MFMA roof).  Its transpose
// dW^T [Fop][Kp] = d_hw^T X is the GAT layers' shape -- 128 x 160 tiles with LDS-direct operand copies (gemm_tn_lds_kernel) -- whenever Kp
// splits into 160-column tiles without more padding than 64-column ones; the slice reduction writes it back transposed.
// Returns the number of k-slices of that route, 0 = not eligible.
static int gcn_dwt_splits(int n, int Kp, int Fop) {
    if (n < 4096 || ((Kp + 159) / 160) * 160 > ((Kp + 63) / 64) * 64) return 0;
    const int tiles = ((Fop + 127) / 128) * ((Kp + 159) / 160);
    const int slots = 2 * device_cu_count();
    int S = slots / tiles;
    const int nkt = (n + GEMM_BK - 1) / GEMM_BK;
    if (S > nkt / 8) S = nkt / 8;                                   // >= 8 k-tiles per slice
    if (S > 64) S = 64;
    return S >= 2 ? S : 0;
}
// out[k][f] = sum_s part[s][f][k]   (k < rows, f < cols; part slices [Fop][ldp]; slices added in order).  A workgroup owns 8 f x 32 k
// elements, one per thread, sixteen slices' loads in flight (the first version -- 32 x 32 tiles, one load in flight -- took 72 us for 42 MB).
__global__ __launch_bounds__(256) void reduce_splits_transposed_kernel(const float* __restrict__ part, int S, long long stride, int rows, int cols,
                                                                        int ldp, float* __restrict__ out) {
    __shared__ float tile[8][33];
    const int f0 = blockIdx.x * 8, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int f = min(f0 + ty, cols - 1), k = min(k0 + tx, rows - 1);
    const float* p = part + (long long)f * ldp + k;
    float acc = 0.f;
    for (int s0 = 0; s0 < S; s0 += 16) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = p[(long long)min(s0 + q, S - 1) * stride];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += (s0 + q < S) ? v[q] : 0.f;
    }
    tile[ty][tx] = acc;
    __syncthreads();
    const int kk = threadIdx.x >> 3, ff = threadIdx.x & 7;          // 32 k rows x 8 f: a row's 8 floats are consecutive in `out`
    if (k0 + kk < rows && f0 + ff < cols) out[(long long)(k0 + kk) * cols + f0 + ff] = tile[ff][kk];
}

}  // namespace txe
using namespace txe;
extern "C" {

int txe_gat_padded_k(int Kh, int Pd) { return round_up(Kh + Pd, 32); }
int txe_gat_padded_f(int H, int D) { return round_up(H * D + 2 * H, 128); }

// Wp [Fp][Kp] from fc.weight W [H*D][Kt], attn_l / attn_r [H*D]   (model_zoo.py:56,65-66)
int txe_gat_pack_weights(const float* W, const float* attn_l, const float* attn_r, int H, int D, int Kt, float* Wp, void* stream) {
    if (!W || !attn_l || !attn_r || !Wp || H < 1 || D < 1 || Kt < 1) return TXE_ERR_ARG;
    const int F = H * D, Fe = F + 2 * H, Fp = round_up(Fe, 128), Kp = round_up(Kt, 32);
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)Fp * Kp;
    hipLaunchKernelGGL(pack_w_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, W, F, Fe, Fp, Kt, Kp, Wp);
    hipLaunchKernelGGL(fold_attn_kernel, dim3((Kt + 63) / 64, 2 * H), dim3(64 * FOLD_DG), 0, s, W, (long long)Kt, Kt, attn_l, attn_r, H, D,
                       Wp + (long long)F * Kp, (long long)Kp);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// Layer input in padded layout (model_zoo.py:214-215 `cat(h, Emb[pos])`).  h == NULL: the feature part [0, Kh) is already in
// place (written by the previous layer's aggregation), only the position-embedding and padding columns are filled.
int txe_gat_build_x(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd, float* X, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || !X || (Pd > 0 && (!pos || !P))) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    const int Kp = round_up(Kh + Pd, 32);
    const long long n = (long long)n_nodes * (Kp - (h ? 0 : Kh));
    if (n == 0) return TXE_OK;
    hipLaunchKernelGGL(build_x_kernel, dim3((int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, h,
                       ld_h, pos, P, n_nodes, Kh, Pd, Kp, X);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// txe_gat_build_x + txe_gat_pack_weights + txe_dropout_mask (over the [n_nodes][Kh+Pd] layer input; feat_drop_p == 0: mask may be NULL)
// as ONE launch -- the per-layer preparation of GATLayer.forward (model_zoo.py:80-85) costs one dispatch instead of four.
int txe_gat_layer_prepare(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd, float* X,
                          const float* W, const float* attn_l, const float* attn_r, int H, int D, float* Wp, float feat_drop_p,
                          unsigned long long seed, unsigned* mask, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || !X || (Pd > 0 && (!pos || !P)) || !W || !attn_l || !attn_r || !Wp || H < 1 || D < 1)
        return TXE_ERR_ARG;
    if (feat_drop_p < 0.f || feat_drop_p >= 1.f || (feat_drop_p > 0.f && !mask)) return TXE_ERR_ARG;
    const int T = 64 * FOLD_DG;
    PrepArgs a;
    memset(&a, 0, sizeof(a));
    a.Kt = Kh + Pd; a.Kp = round_up(a.Kt, 32);
    a.F = H * D; a.Fe = a.F + 2 * H; a.Fp = round_up(a.Fe, 128);
    auto blocks = [&](long long n, int cap) { const long long b = (n + T - 1) / T; return (int)(b < cap ? b : cap); };
    const long long nx = (long long)n_nodes * (a.Kp - (h ? 0 : Kh));
    a.nb_x = nx > 0 ? build_x_blocks(T, n_nodes, Kh, Pd, a.Kp, h != nullptr, false, 2048) : 0;
    a.n_words = (feat_drop_p > 0.f) ? (long long)n_nodes * ((a.Kt + 31) / 32) : 0;
    a.nb_m = blocks(a.n_words, 1024);
    a.nb_w = blocks(((long long)a.Fp * a.Kp / 4 + PREP_U - 1) / PREP_U, 512);   // PREP_U quads per thread
    a.fold_bx = (a.Kt + 63) / 64;
    a.nb_f = a.fold_bx * 2 * H;
    a.h = h; a.ld_h = ld_h; a.pos = pos; a.P = P; a.n_rows = n_nodes; a.Kh = Kh; a.Pd = Pd; a.X = X;
    a.W = W; a.attn_l = attn_l; a.attn_r = attn_r; a.H = H; a.D = D; a.Wp = Wp;
    a.pk_rows = a.F; a.pk_ext = a.Fe; a.pk_prows = a.Fp; a.pk_cols = a.Kt; a.pk_pcols = a.Kp;
    a.seed = seed; a.thr16 = (unsigned)(feat_drop_p * 65536.0f + 0.5f); a.mask = mask;
    a.x_dropped = 0; a.drop_scale = 1.f;                                  // (this entry leaves the dropout to the GEMM loaders)
    hipLaunchKernelGGL(gat_prepare_kernel, dim3(a.nb_x + a.nb_m + a.nb_w + a.nb_f), dim3(T), 0, (hipStream_t)stream, a);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// txe_gat_layer_prepare for ALL GATLayers of a stack in ONE launch: everything a layer needs before its projection -- packed weights,
// folded attention rows, keep mask, the position-embedding / padding columns of its input -- depends on the parameters and on `pos`
// only, never on the layer below's output, so the whole stack can be prepared before the first GEMM (one dispatch instead of one per
// layer; the feature columns of the deeper layers' inputs are written later by the aggregation below them).
struct txe_gat_prepare_desc {
    const float* h; long long ld_h; int n_nodes, Kh; const int* pos; const float* P; int Pd; float* X;
    const float *W, *attn_l, *attn_r; int H, D; float* Wp; float feat_drop_p; unsigned long long seed; unsigned* mask;
    int x_dropped;
};
}  // extern "C"
namespace txe {
constexpr int PREP_MAXL = 4;
struct PrepMulti {
    int n; int nb_end[PREP_MAXL]; PrepArgs a[PREP_MAXL];
    int nb_norm; const int* norm_rowptr; int norm_n; float* norm;   // GCN stacks: norm = in_degree^-1/2 (model_zoo.py:157-161) by leading workgroups
};
__global__ __launch_bounds__(64 * FOLD_DG) void gat_prepare_multi_kernel(const PrepMulti m) {
    int b = blockIdx.x, i = 0;
    if (b < m.nb_norm) {
        const int v = b * (64 * FOLD_DG) + threadIdx.x;
        if (v < m.norm_n) { const int deg = m.norm_rowptr[v + 1] - m.norm_rowptr[v]; m.norm[v] = deg > 0 ? 1.0f / sqrtf((float)deg) : 0.f; }
        return;
    }
    b -= m.nb_norm;
    while (i + 1 < m.n && b >= m.nb_end[i]) ++i;                    // (block-uniform)
    prepare_jobs(m.a[i], b - ((i > 0) ? m.nb_end[i - 1] : 0));
}
// (measured on the 18 k-node training batch, 35 us layer after layer: the VALU-bound mask jobs and the streaming jobs dealt
//  alternately, one of each per CU: 40.5 us; JOB-major order over the layers -- folds, build_x, packs, masks: 35.9 us, and 150
//  against 138 us on the 1.1 M-node inference batch; the deeper layers' preparation on the second stream under the first
//  projection GEMM: the step unchanged -- its workgroups crawl beside the persistent GEMM's and that GEMM loses what was gained)
static int fill_prep(PrepArgs& a, const txe_gat_prepare_desc& d) {
    // (X == NULL: the layer's input is not stored -- txe_gat_dense_fwd_split_src forms it from h; mask and weights are still written)
    if (d.n_nodes < 0 || d.Kh < 1 || d.Pd < 0 || (!d.X && !d.h) || (d.Pd > 0 && (!d.pos || !d.P)) || !d.W || !d.attn_l || !d.attn_r || !d.Wp || d.H < 1 ||
        d.D < 1 || d.feat_drop_p < 0.f || d.feat_drop_p >= 1.f || (d.feat_drop_p > 0.f && !d.mask))
        return TXE_ERR_ARG;
    const int T = 64 * FOLD_DG;
    a.Kt = d.Kh + d.Pd; a.Kp = round_up(a.Kt, 32);
    a.F = d.H * d.D; a.Fe = a.F + 2 * d.H; a.Fp = round_up(a.Fe, 128);
    auto blocks = [&](long long n, int cap) { const long long b = (n + T - 1) / T; return (int)(b < cap ? b : cap); };
    const long long nx = d.X ? (long long)d.n_nodes * (a.Kp - (d.h ? 0 : d.Kh)) : 0;
    a.seed = d.seed; a.thr16 = (unsigned)(d.feat_drop_p * 65536.0f + 0.5f); a.mask = d.mask;
    a.x_dropped = (d.x_dropped && d.feat_drop_p > 0.f && a.thr16 != 0u) ? 1 : 0;
    a.x_mask = (a.x_dropped && d.h != nullptr && nx > 0) ? 1 : 0;     // build_x hashes every word of its rows anyway: it writes the mask
    a.drop_scale = 1.f / (1.f - d.feat_drop_p);
    a.nb_x = nx > 0 ? build_x_blocks(T, d.n_nodes, d.Kh, d.Pd, a.Kp, d.h != nullptr, a.x_dropped != 0, 2048) : 0;
    a.n_words = (d.feat_drop_p > 0.f) ? (long long)d.n_nodes * ((a.Kt + 31) / 32) : 0;
    a.nb_m = a.x_mask ? 0 : blocks(a.n_words, 1024);
    a.nb_w = blocks(((long long)a.Fp * a.Kp / 4 + PREP_U - 1) / PREP_U, 512);
    a.fold_bx = (a.Kt + 63) / 64;
    a.nb_f = a.fold_bx * 2 * d.H;
    a.h = d.h; a.ld_h = d.ld_h; a.pos = d.pos; a.P = d.P; a.n_rows = d.n_nodes; a.Kh = d.Kh; a.Pd = d.Pd; a.X = d.X;
    a.W = d.W; a.attn_l = d.attn_l; a.attn_r = d.attn_r; a.H = d.H; a.D = d.D; a.Wp = d.Wp;
    a.pk_rows = a.F; a.pk_ext = a.Fe; a.pk_prows = a.Fp; a.pk_cols = a.Kt; a.pk_pcols = a.Kp;
    return TXE_OK;
}
}  // namespace txe
extern "C" {
int txe_gat_layers_prepare(const struct txe_gat_prepare_desc* descs, int n_layers, void* stream) {
    if (!descs || n_layers < 1) return TXE_ERR_ARG;
    for (int i0 = 0; i0 < n_layers; i0 += PREP_MAXL) {              // (more than PREP_MAXL layers: several launches)
        PrepMulti m;
        memset(&m, 0, sizeof(m));
        m.n = n_layers - i0 < PREP_MAXL ? n_layers - i0 : PREP_MAXL;
        int total = 0;
        for (int i = 0; i < m.n; ++i) {
            const int rc = fill_prep(m.a[i], descs[i0 + i]);
            if (rc) return rc;
            total += m.a[i].nb_x + m.a[i].nb_m + m.a[i].nb_w + m.a[i].nb_f;
            m.nb_end[i] = total;
        }
        if (total == 0) continue;
        hipLaunchKernelGGL(gat_prepare_multi_kernel, dim3(total), dim3(64 * FOLD_DG), 0, (hipStream_t)stream, m);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

// The same for a GCNLayer (model_zoo.py:35-37): txe_gat_build_x + txe_gcn_pack_weights + txe_dropout_mask in one launch.
// W [Kh+Pd][Fo] -> Wp [roundup(roundup(Kh+Pd,32),128)][roundup(Fo,32)]; mask may be NULL when drop_p == 0.
struct txe_gcn_prepare_desc {
    const float* h; long long ld_h; int n_nodes, Kh; const int* pos; const float* P; int Pd; float* X;
    const float* W; int Fo; float* Wp; float drop_p; unsigned long long seed; unsigned* mask; int x_dropped; const float* bias_row;
};
}  // extern "C"
namespace txe {
static int fill_prep_gcn(PrepArgs& a, const txe_gcn_prepare_desc& d) {
    // bias_row (or NULL): packed as row Kh + Pd of Wp (needs a padding row: (Kh + Pd) % 32 != 0) -- the folded output layer then carries
    // its bias as the weight row of a column of Z that counts as 1 (txe_bilinear_folded_*: one_col)
    if (d.n_nodes < 0 || d.Kh < 1 || d.Pd < 0 || !d.X || (d.Pd > 0 && (!d.pos || !d.P)) || !d.W || !d.Wp || d.Fo < 1) return TXE_ERR_ARG;
    if (d.bias_row && ((d.Kh + d.Pd) % 32) == 0) return TXE_ERR_ARG;
    if (d.drop_p < 0.f || d.drop_p >= 1.f || (d.drop_p > 0.f && !d.mask)) return TXE_ERR_ARG;
    const int T = 64 * FOLD_DG;
    memset(&a, 0, sizeof(a));
    a.Kt = d.Kh + d.Pd; a.Kp = round_up(a.Kt, 32);
    auto blocks = [&](long long n, int cap) { const long long b = (n + T - 1) / T; return (int)(b < cap ? b : cap); };
    const long long nx = (long long)d.n_nodes * (a.Kp - (d.h ? 0 : d.Kh));
    a.seed = d.seed; a.thr16 = (unsigned)(d.drop_p * 65536.0f + 0.5f); a.mask = d.mask;
    a.x_dropped = (d.x_dropped && d.drop_p > 0.f && a.thr16 != 0u) ? 1 : 0;      // (as txe_gat_prepare_desc.x_dropped)
    a.x_mask = (a.x_dropped && d.h != nullptr && nx > 0) ? 1 : 0;
    a.drop_scale = 1.f / (1.f - d.drop_p);
    a.nb_x = nx > 0 ? build_x_blocks(T, d.n_nodes, d.Kh, d.Pd, a.Kp, d.h != nullptr, a.x_dropped != 0, 2048) : 0;
    a.n_words = (d.drop_p > 0.f) ? (long long)d.n_nodes * ((a.Kt + 31) / 32) : 0;
    a.nb_m = a.x_mask ? 0 : blocks(a.n_words, 1024);
    a.pk_rows = a.Kt; a.pk_ext = a.Kt; a.pk_prows = round_up(a.Kp, 128); a.pk_cols = d.Fo; a.pk_pcols = round_up(d.Fo, 32);
    a.nb_w = blocks(((long long)a.pk_prows * a.pk_pcols / 4 + PREP_U - 1) / PREP_U, 512);
    a.nb_f = 0; a.fold_bx = 1;
    a.h = d.h; a.ld_h = d.ld_h; a.pos = d.pos; a.P = d.P; a.n_rows = d.n_nodes; a.Kh = d.Kh; a.Pd = d.Pd; a.X = d.X;
    a.W = d.W; a.Wp = d.Wp; a.pk_extra = d.bias_row;
    return TXE_OK;
}
}  // namespace txe
extern "C" {
int txe_gcn_layer_prepare(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd, float* X,
                          const float* W, int Fo, float* Wp, float drop_p, unsigned long long seed, unsigned* mask, int x_dropped,
                          const float* bias_row, void* stream) {
    const txe_gcn_prepare_desc d{h, ld_h, n_nodes, Kh, pos, P, Pd, X, W, Fo, Wp, drop_p, seed, mask, x_dropped, bias_row};
    PrepArgs a;
    const int rc = fill_prep_gcn(a, d);
    if (rc) return rc;
    hipLaunchKernelGGL(gat_prepare_kernel, dim3(a.nb_x + a.nb_m + a.nb_w), dim3(64 * FOLD_DG), 0, (hipStream_t)stream, a);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// ... for every GCNLayer of a stack in ONE launch (a layer's preparation never depends on the layer below's output), together with the
// stack's degree normalisation norm[v] = in_degree(v)^-1/2 (txe_gcn_norm; rowptr_in == NULL: without it)
int txe_gcn_layers_prepare(const struct txe_gcn_prepare_desc* descs, int n_layers, const int* rowptr_in, int n_nodes, float* norm, void* stream) {
    if (!descs || n_layers < 1 || (rowptr_in && (n_nodes < 0 || !norm))) return TXE_ERR_ARG;
    for (int i0 = 0; i0 < n_layers; i0 += PREP_MAXL) {
        PrepMulti m;
        memset(&m, 0, sizeof(m));
        m.n = n_layers - i0 < PREP_MAXL ? n_layers - i0 : PREP_MAXL;
        if (i0 == 0 && rowptr_in && n_nodes > 0) {
            m.nb_norm = (n_nodes + 64 * FOLD_DG - 1) / (64 * FOLD_DG); m.norm_rowptr = rowptr_in; m.norm_n = n_nodes; m.norm = norm;
        }
        int total = 0;
        for (int i = 0; i < m.n; ++i) {
            const int rc = fill_prep_gcn(m.a[i], descs[i0 + i]);
            if (rc) return rc;
            total += m.a[i].nb_x + m.a[i].nb_m + m.a[i].nb_w;
            m.nb_end[i] = total;
        }
        if (total + m.nb_norm == 0) continue;
        hipLaunchKernelGGL(gat_prepare_multi_kernel, dim3(total + m.nb_norm), dim3(64 * FOLD_DG), 0, (hipStream_t)stream, m);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

// Eval-mode layer-0 projection of a batch drawn from a feature table (SURVEY 8f-2 "dedup by _id"): Y[v] = T[row[v]] + T2[row2[v]],
// n_cols a multiple of 4, 16-byte aligned rows.  T2 / row2 may be NULL.
int txe_gather_add_rows(const float* T, long long ld_t, const int* row, const float* T2, long long ld_t2, const int* row2, long long n_rows,
                        int n_cols, float* Y, long long ld_y, void* stream) {
    if (n_rows < 0 || n_cols < 4 || (n_cols & 3) || !T || !row || !Y || (T2 && !row2) || (ld_t & 3) || (ld_y & 3) || (T2 && (ld_t2 & 3)))
        return TXE_ERR_ARG;
    if ((((uintptr_t)T | (uintptr_t)Y | (uintptr_t)T2) & 15) != 0) return TXE_ERR_ARG;
    if (n_rows == 0) return TXE_OK;
    const int nvec = n_cols / 4;
    const long long total = n_rows * nvec;
    const int nb = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ProfScope prof("gather_add_rows_kernel", (hipStream_t)stream, 4.0 * 2.0 * n_rows * (double)n_cols, 1);     // read a row, write a row
    hipLaunchKernelGGL(gather_add_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, T, ld_t, row, T2, ld_t2, row2, n_rows, nvec, Y, ld_y);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// 1: txe_gat_dense_bwd forms this layer's d_X with the streaming position-column kernel (phase 1 is then an HBM stream that belongs
// IN LINE on the caller's stream, not beside the weight-gradient product on a second one)
int txe_gat_dx_streams(int Kh, int Pd, int need_dh) {
    if (need_dh || Pd < 1 || Kh < 1) return 0;
    const int c0 = (Kh / 4) * 4;
    return (Kh + Pd - c0 <= DXPOS_MAXC && Kh - c0 + Pd <= DXPOS_MAXC) ? 1 : 0;
}

// extra workspace (behind txe_gat_dense_ws_bytes) with which txe_gat_dense_bwd forms a need_dh layer's d_X on the bf16 pipe
static inline size_t dense_bwd_split_bytes(int n_nodes, int Fp, int Kt) {
    return align_up(split_packed_bytes(n_nodes, Fp), 256) + align_up(split_packed_bytes(Kt, Fp), 256);
}
size_t txe_gat_dense_bwd_split_ws_bytes(int n_nodes, int Kh, int Pd, int H, int D) {
    if (n_nodes < 1 || Kh < 1 || Pd < 0 || H < 1 || D < 1) return 0;
    return dense_bwd_split_bytes(n_nodes, round_up(H * D + 2 * H, 128), Kh + Pd);
}
size_t txe_gat_dense_ws_bytes(int n_nodes, int Kh, int Pd, int H, int D, int vocab) {
    return plan_dense_ws(nullptr, n_nodes, round_up(H * D + 2 * H, 128), 2 * H, round_up(Kh + Pd, 32), Pd, vocab).total;
}

// Y [N][Fp] = dropout(X) [N][Kp] * Wp^T      (model_zoo.py:82-85: feat_drop, fc, a1, a2 in one product)
int txe_gat_dense_fwd(const float* X, int n_nodes, int Kh, int Pd, const float* Wp, int H, int D, float feat_drop_p,
                      const unsigned* mask, float* Y, void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || H < 1 || D < 1 || !X || !Wp || !Y) return TXE_ERR_ARG;
    if (feat_drop_p < 0.f || feat_drop_p >= 1.f) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    const int Fe = H * D + 2 * H, Fp = round_up(Fe, 128), Kp = round_up(Kh + Pd, 32);
    VMat A = vmat_plain(X, Kp, n_nodes, Kp);
    vmat_set_mask(A, mask, feat_drop_p);
    VMat B = vmat_plain(Wp, Kp, Fp, Kp);
    Epi E = epi_plain(Y, Fp, Fe);
    E.alg_flops = 2.0 * n_nodes * (double)Fe * (Kh + Pd);           // without the k-tile padding of X / Wp
    E.k_valid = Kh + Pd;                                             // (X and Wp carry zeros behind it: txe_gat_layers_prepare / pack_w)
    const bool tail_ok = ws && ws_bytes >= gemm_tail_ws_bytes();
    return gemm_nt(A, B, E, n_nodes, Fe, Kp, 1, (hipStream_t)stream, tail_ok ? ws : nullptr, tail_ok ? ws_bytes : 0);
}

// The same product on the bf16 matrix pipe (txe_gemm_split.h: three bf16 planes per fp32 operand, six plane products, fp32
// accumulation -- fp32 accuracy at 6/16 of the fp32 MFMA's time).  X is a PLAIN operand here: dropout(X) already applied
// (txe_gat_prepare_desc.x_dropped) or no dropout.  Xs / Ws: the packed planes of X (side 0) / Wp (side 1) when the preparation launch
// wrote them, else NULL -- they are then packed here, into ws.  Xt_out (or NULL): txe_gat_dense_split_xt_bytes for X packed
// contraction-major, what txe_gat_dense_bwd's weight gradient takes on the same pipe.
size_t txe_gat_dense_split_ws_bytes(int n_nodes, int Kh, int Pd, int H, int D) {
    if (n_nodes < 1 || Kh < 1 || Pd < 0 || H < 1 || D < 1) return 0;
    const int Fp = round_up(H * D + 2 * H, 128), Kp = round_up(Kh + Pd, 32);
    return align_up(split_packed_bytes(n_nodes, Kp), 256) + align_up(split_packed_bytes(Fp, Kp), 256);
}
size_t txe_gat_dense_split_xt_bytes(int n_nodes, int Kh, int Pd, int H, int D) {       // 0: this layer's weight gradient keeps the fp32 route
    if (n_nodes < 1 || Kh < 1 || Pd < 0 || H < 1 || D < 1) return 0;
    const int Fp = round_up(H * D + 2 * H, 128), Kp = round_up(Kh + Pd, 32);
    return (split_tn_eligible(Fp, Kp) && split_tn_fits(n_nodes, Fp)) ? split_packed_t_bytes(n_nodes, Kp) : 0;
}
int txe_gat_dense_fwd_split(const float* X, int n_nodes, int Kh, int Pd, const float* Wp, int H, int D, const void* Xs, const void* Ws,
                            void* Xt_out, float* Y, void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || H < 1 || D < 1 || !Y || (!Xs && !X) || (!Ws && !Wp)) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    const int Fe = H * D + 2 * H, Fp = round_up(Fe, 128), Kp = round_up(Kh + Pd, 32);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    size_t off = 0;
    int rc;
    if (Xt_out && (!X || !split_tn_eligible(Fp, Kp))) return TXE_ERR_ARG;
    bool xt_done = false;
    // the contraction runs over whole k-tiles of 16: Kc columns (X and Wp hold zeros in [Kh + Pd, Kp))
    const int Kc = round_up(Kh + Pd, 16);
    if (!Xs && !Ws) {                                   // the usual case: all three packs in one launch
        const size_t ba = align_up(split_packed_bytes(n_nodes, Kp), 256), bb = align_up(split_packed_bytes(Fp, Kp), 256);
        if (!ws || ws_bytes < ba + bb) return TXE_ERR_WORKSPACE;
        rc = split_pack_layer_launch(X, Kp, n_nodes, Wp, Kp, Fp, Kc, Kp, w, w + ba, Xt_out, s);
        if (rc) return rc;
        Xs = w; Ws = w + ba; xt_done = true;
    }
    if (!Xs) {
        const size_t b = align_up(split_packed_bytes(n_nodes, Kp), 256);
        if (!ws || ws_bytes < off + b) return TXE_ERR_WORKSPACE;
        rc = split_pack_launch(X, Kp, n_nodes, Kc, 0, w + off, s);
        if (rc) return rc;
        Xs = w + off; off += b;
    }
    if (!Ws) {
        const size_t b = align_up(split_packed_bytes(Fp, Kp), 256);
        if (!ws || ws_bytes < off + b) return TXE_ERR_WORKSPACE;
        rc = split_pack_launch(Wp, Kp, Fp, Kc, 1, w + off, s);
        if (rc) return rc;
        Ws = w + off; off += b;
    }
    rc = gemm_nt_split_launch(Xs, Ws, n_nodes, Fe, Kc, Y, Fp, 2.0 * n_nodes * (double)Fe * (Kh + Pd), s);
    if (rc) return rc;
    // (X packed contraction-major for the backward pass's weight gradient, txe_gat_dense_bwd: Xt)
    if (Xt_out && !xt_done) return split_pack_t_launch(X, Kp, n_nodes, Kp, Xt_out, s);
    return TXE_OK;
}

// The same product for a FIRST layer whose input X = dropout([h | Emb[pos]]) is never stored: the packs form its elements from h, the
// position table and the keep mask (txe_gat_layers_prepare with X == NULL writes mask and weights only) -- one 23-MB write and two reads
// of it less per step on the training batch.  mask == NULL or feat_drop_p == 0: no dropout.
int txe_gat_dense_fwd_split_src(const float* h, long long ld_h, const int* pos, const float* P, const unsigned* mask, float feat_drop_p,
                                int n_nodes, int Kh, int Pd, const float* Wp, int H, int D, void* Xt_out, float* Y, void* ws, size_t ws_bytes,
                                void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || H < 1 || D < 1 || !Y || !h || !Wp || ld_h < Kh || (Pd > 0 && (!pos || !P)) || feat_drop_p < 0.f ||
        feat_drop_p >= 1.f || (feat_drop_p > 0.f && !mask))
        return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    const int Fe = H * D + 2 * H, Fp = round_up(Fe, 128), Kp = round_up(Kh + Pd, 32);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    if (Xt_out && !split_tn_eligible(Fp, Kp)) return TXE_ERR_ARG;
    const int Kc = round_up(Kh + Pd, 16);
    const size_t ba = align_up(split_packed_bytes(n_nodes, Kp), 256), bb = align_up(split_packed_bytes(Fp, Kp), 256);
    if (!ws || ws_bytes < ba + bb) return TXE_ERR_WORKSPACE;
    SplitVSrc vs{h, ld_h, pos, P, Kh, Pd, feat_drop_p > 0.f ? mask : nullptr, (Kh + Pd + 31) / 32, 1.f / (1.f - feat_drop_p)};
    int rc = split_pack_layer_launch(nullptr, Kp, n_nodes, Wp, Kp, Fp, Kc, Kp, w, w + ba, Xt_out, s, &vs);
    if (rc) return rc;
    return gemm_nt_split_launch(w, w + ba, n_nodes, Fe, Kc, Y, Fp, 2.0 * n_nodes * (double)Fe * (Kh + Pd), s);
}

// Backward of txe_gat_dense_fwd.  d_Y [N][Fp] must have ZERO padding columns [F+2H, Fp).
//   d_X [N][Kp]: columns [c0, Kt) are written, c0 = 0 if need_dh else the 32-aligned start of the position columns;
//                columns < Kh are multiplied by leaky'(X) when act_slope_on (X[:, :Kh] is then the activated output of the
//                previous layer), all by the dropout factor.
//   dW [F][Kt], d_attn_l / d_attn_r [F], dP [vocab][Pd].
// phases: 7 = everything; 1 = d_X, 2 = the dW GEMM (independent of each other), 4 = the reductions that need both -- separate calls
// share the workspace.  phases | 16: the full d_X product (need_dh) runs on the bf16 matrix pipe, its packed operands behind the workspace
// (ws_bytes >= txe_gat_dense_ws_bytes + txe_gat_dense_bwd_split_ws_bytes, else TXE_ERR_WORKSPACE); without the bit: the fp32 MFMA.
int txe_gat_dense_bwd(const float* X, int n_nodes, int Kh, int Pd, const int* pos, int vocab, const float* Wp, const float* W,
                      const float* attn_l, const float* attn_r, int H, int D, float feat_drop_p, const unsigned* mask, const float* d_Y,
                      int need_dh, int act_on, float act_slope, float* d_X, float* dW, float* d_attn_l, float* d_attn_r, float* dP,
                      int x_dropped, const void* Xt, int phases, void* chain, void* ws, size_t ws_bytes, void* stream) {
    // (X == NULL: a first layer whose input was never stored, txe_gat_dense_fwd_split_src -- its weight gradient needs Xt then)
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || H < 1 || D < 1 || (!X && (!Xt || act_on)) || !Wp || !W || !attn_l || !attn_r || !d_Y || !dW || !d_attn_l || !d_attn_r || !ws)
        return TXE_ERR_ARG;
    static_assert(sizeof(TailChain) <= TXE_TAIL_CHAIN_BYTES, "txe.h: TXE_TAIL_CHAIN_BYTES");
    if ((need_dh || Pd > 0) && !d_X) return TXE_ERR_ARG;
    if (Pd > 0 && (!pos || !dP || vocab < 1 || vocab > MAX_VOCAB)) return TXE_ERR_ARG;
    if (feat_drop_p < 0.f || feat_drop_p >= 1.f) return TXE_ERR_ARG;
    const int F = H * D, H2 = 2 * H, Fe = F + H2, Fp = round_up(Fe, 128), Kt = Kh + Pd, Kp = round_up(Kt, 32);
    DenseWs p = plan_dense_ws(ws, n_nodes, Fp, H2, Kp, Pd, vocab);
    if (ws_bytes < p.total) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // ---- d_X[:, c0:Kt] = d_Y * Wp[:, c0:Kt] ----
    const int c0 = need_dh ? 0 : (Kh / 4) * 4;      // 16-byte aligned start of the position columns
    // position columns only: one stream over d_Y (txe_dxpos.hip) that also leaves the per-class partial sums of dP
    const bool stream_dx = txe_gat_dx_streams(Kh, Pd, need_dh) == 1 && n_nodes > 0;
    DxPosArgs da;
    memset(&da, 0, sizeof(da));
    if (stream_dx) {
        da.dY = d_Y; da.ld_dy = Fp; da.n_rows = n_nodes; da.K = Fp;
        da.Wp = Wp; da.ld_w = Kp; da.Kp = Kp; da.c0 = c0; da.NC = Kt - c0;
        da.mask = mask; da.mask_ld = (Kt + 31) / 32; da.mask_on = (mask && feat_drop_p > 0.f) ? 1 : 0;
        da.drop_scale = da.mask_on ? 1.f / (1.f - feat_drop_p) : 1.f;
        da.dX = d_X; da.ld_dx = Kp;
        da.pos = pos; da.vocab = vocab; da.Pd = Pd; da.pcol0 = Kh - c0; da.ppart = p.ppart; da.part = p.dxpart;
        rc = dxpos_prepare(da);
        if (rc) return rc;
    }
    if ((phases & 1) && stream_dx) {
        rc = dxpos_launch(da, s);
        if (rc) return rc;
    } else if ((phases & 1) && (phases & 16) && need_dh && n_nodes > 0) {
        if (ws_bytes < p.total + dense_bwd_split_bytes(n_nodes, Fp, Kt)) return TXE_ERR_WORKSPACE;     // (the route is the caller's choice, not the buffer's size)
        // the whole d_X = d_Y Wp on the bf16 pipe (txe_gemm_split.h): d_Y packed as the row operand, Wp -- given as the transpose of
        // the column operand -- packed from its columns; dropout mask and leaky' factor in the store loop (epi_store_one's arithmetic)
        char* sw = (char*)ws + p.total;
        const size_t ba = align_up(split_packed_bytes(n_nodes, Fp), 256);
        const int Fc = round_up(Fe, 16);                 // whole k-tiles of 16 over the contraction (d_Y's columns past Fe are zeros)
        rc = split_pack_launch(d_Y, Fp, n_nodes, Fc, 0, sw, s);
        if (rc) return rc;
        rc = split_pack_launch(Wp, Kp, Kt, Fc, 3, sw + ba, s);
        if (rc) return rc;
        SplitEpi e;
        memset(&e, 0, sizeof(e));
        e.drop_scale = 1.f;
        if (mask && feat_drop_p > 0.f) { e.mask = mask; e.mask_ld = (Kt + 31) / 32; e.mask_col0 = 0; e.drop_scale = 1.f / (1.f - feat_drop_p); }
        if (act_on) { e.act_src = X; e.ld_act = Kp; e.act_slope = act_slope; e.cols_act = Kh; }
        rc = gemm_nt_split_launch(sw, sw + ba, n_nodes, Kt, Fc, d_X, Kp, 2.0 * n_nodes * (double)Kt * Fe, s, &e);
        if (rc) return rc;
    } else if ((phases & 1) && Kt - c0 > 0 && n_nodes > 0 && (need_dh || Pd > 0)) {
        VMat A = vmat_plain(d_Y, Fp, n_nodes, Fp);
        VMat B = vmat_plain(Wp + c0, Kp, Fp, Kp - c0);
        Epi E = epi_plain(d_X + c0, Kp, Kh > c0 ? Kh - c0 : 0);
        E.c2 = d_X + c0 + E.cols_main; E.ldc2 = Kp;                 // same buffer: the split only scopes the activation factor
        epi_set_mask(E, mask, Kt, c0, feat_drop_p);
        if (act_on && need_dh) epi_set_act(E, X + c0, Kp, act_slope);
        E.alg_flops = 2.0 * n_nodes * (double)(need_dh ? Kt : Pd) * Fe;
        rc = gemm_nn(A, B, E, n_nodes, Kt - c0, Fp, 1, s, p.tail, p.tail_bytes);
        if (rc) return rc;
    }
    // ---- dWp = d_Y^T * dropout(X)  (split-K over the node dimension) ----
    VMat A = vmat_plain(d_Y, Fp, n_nodes, Fp);
    VMat B = vmat_plain(X, Kp, n_nodes, Kp);
    if (!x_dropped) vmat_set_mask(B, mask, feat_drop_p);            // (x_dropped: X already holds dropout(X), txe_gat_layers_prepare)
    Epi E = epi_plain(p.part, Kp, Kp);
    E.split_stride = (long long)Fp * Kp;
    E.alg_flops = 2.0 * Fe * (double)Kt * n_nodes;
    const int splits = p.splits;
    if ((phases & 2) && Xt && x_dropped + (feat_drop_p == 0.f) > 0 && split_tn_eligible(Fp, Kp) && split_tn_fits(n_nodes, Fp) && n_nodes > 0) {
        // the same slices on the bf16 pipe (txe_gemm_split.h): X packed contraction-major by the forward pass, d_Y split in the loader
        const int ks = round_up((n_nodes + splits - 1) / splits, 16);
        rc = gemm_tn_split_launch(d_Y, Fp, Fp, Xt, Kp, n_nodes, splits, ks, p.part, Kp, E.split_stride, E.alg_flops, s);
        if (rc) return rc;
    } else if (phases & 2) {
        if (!X) return TXE_ERR_ARG;
        rc = gemm_tn(A, B, E, Fp, Kp, n_nodes, splits, s);
        if (rc) return rc;
    }
    if (!(phases & 4)) return TXE_OK;
    const int S = n_nodes > 0 ? splits : 0;
    // ---- phase A: dP partials (dP[c][j] = sum_{pos[m]==c} d_X[m][Kh+j]) and d_wa = the extension rows of dWp ----
    const int nseg = (Pd > 0 && n_nodes > 0) ? (stream_dx ? dxpos_blocks(n_nodes) : p.seg_blocks) : 0;
    TailA ta;
    memset(&ta, 0, sizeof(ta));
    ta.nb_dx = stream_dx ? nseg : 0; ta.dx = da;
    ta.nb_s1a = stream_dx ? 0 : nseg; ta.s1a = Seg1Args{d_X ? d_X + Kh : nullptr, (long long)Kp, Pd, p.ppart};
    ta.pos = pos; ta.n_rows = n_nodes; ta.vocab = vocab; ta.rows_per_block = p.seg_rows;
    ta.r_kind = 1; ta.nbx = (Kp + 255) / 256; ta.nb_r = ta.nbx * H2;
    ta.rpart = p.part; ta.S = S; ta.split_stride = E.split_stride; ta.F = F; ta.ldp = Kp; ta.dwa = p.dwa;
    hipLaunchKernelGGL(gat_bwd_reduce_a_kernel, dim3(ta.nb_dx + ta.nb_s1a + ta.nb_r), dim3(256), 0, s, ta);
    TXE_CHECK_LAUNCH();
    // ---- phase B: dW / d_attn (unfold) and dP ----
    TailB tb;
    memset(&tb, 0, sizeof(tb));
    tb.nb_u = F;
    tb.u = UnfoldArgs{p.part, S, E.split_stride, p.dwa, (long long)Kp, W, (long long)Kt, attn_l, attn_r, H, D, Kt, dW, (long long)Kt,
                      d_attn_l, d_attn_r};
    tb.nb_2a = Pd > 0 ? (vocab * Pd + 63) / 64 : 0;
    tb.s2a = Seg2Args{p.ppart, nseg, vocab * Pd, dP};
    return tail_b_submit(&tb, chain, (phases & 64) != 0, s);
}

// launches whatever a chain of deferred phase-B jobs still holds (a stack whose last call deferred too); chain == NULL: nothing
int txe_gat_tail_flush(void* chain, void* stream) {
    if (!chain) return TXE_OK;
    return tail_b_submit(nullptr, chain, false, (hipStream_t)stream);
}

// zero columns [c0, c1) of a row-major fp32 matrix (padding columns of d_Y)
int txe_zero_cols(float* x, long long ld, int n_rows, int c0, int c1, void* stream) {
    if (!x || n_rows < 0 || c0 < 0 || c1 < c0) return TXE_ERR_ARG;
    const long long n = (long long)n_rows * (c1 - c0);
    if (n == 0) return TXE_OK;
    hipLaunchKernelGGL(zero_cols_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, x, ld,
                       n_rows, c0, c1);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}


// ---------------------------------------------------------------------------------------------
// GCNLayer dense part (model_zoo.py:35-37) on padded operands:  hw = dropout(X) Wp,
//   X  [N][Kp]       as for GAT (txe_gat_build_x),
//   Wp [Kp128][Fop]  = weight [Kt][Fo] zero-padded, Kp128 = roundup(Kp,128), Fop = roundup(Fo,32),
//   hw / d_hw [N][Fop]  (d_hw with ZERO padding columns).
// ---------------------------------------------------------------------------------------------
int txe_gcn_padded_f(int Fo) { return round_up(Fo, 32); }

int txe_gcn_pack_weights(const float* W, int Kt, int Fo, float* Wp, void* stream) {
    if (!W || !Wp || Kt < 1 || Fo < 1) return TXE_ERR_ARG;
    const int Kp128 = round_up(round_up(Kt, 32), 128), Fop = round_up(Fo, 32);
    const long long n = (long long)Kp128 * Fop;
    // pack_w_kernel(W, F=rows, Fe=rows, Fp=padded rows, Kt=cols, Kp=padded cols)
    hipLaunchKernelGGL(pack_w_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, W, Kt,
                       Kt, Kp128, Fo, Fop, Wp);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

size_t txe_gcn_dense_ws_bytes(int n_nodes, int Kh, int Pd, int Fo, int vocab) {
    const int Kp = round_up(Kh + Pd, 32), Fop = round_up(Fo, 32);
    return plan_dense_ws(nullptr, n_nodes, Kp, 0, Fop, Pd, vocab, false, gcn_dwt_splits(n_nodes, Kp, Fop)).total;
}

int txe_gcn_dense_fwd(const float* X, int n_nodes, int Kh, int Pd, const float* Wp, int Fo, float drop_p, const unsigned* mask,
                      float* hw, void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || Fo < 1 || !X || !Wp || !hw) return TXE_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    const int Kp = round_up(Kh + Pd, 32), Fop = round_up(Fo, 32);
    VMat A = vmat_plain(X, Kp, n_nodes, Kp);
    vmat_set_mask(A, mask, drop_p);
    VMat B = vmat_plain(Wp, Fop, Kp, Fop);
    Epi E = epi_plain(hw, Fop, Fo);
    E.alg_flops = 2.0 * n_nodes * (double)Fo * (Kh + Pd);
    const bool tail_ok = ws && ws_bytes >= gemm_tail_ws_bytes();
    return gemm_nn(A, B, E, n_nodes, Fo, Kp, 1, (hipStream_t)stream, tail_ok ? ws : nullptr, tail_ok ? ws_bytes : 0);
}

// dW[k][f] = sum_s part[s][k][f]  (k < Kt, f < Fo; part rows have stride ldp)
__global__ void reduce_splits_sub_kernel(const float* __restrict__ part, int S, long long stride, int rows, int cols, int ldp,
                                         float* __restrict__ out) {
    const long long n = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols;
        const int c = (int)(i % cols);
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += part[(long long)s * stride + r * ldp + c];
        out[i] = acc;
    }
}

// d_hw [N][Fop] with zero padding columns.  Writes d_X columns [c0, Kt) (as txe_gat_dense_bwd), dW [Kt][Fo], dP.
int txe_gcn_dense_bwd(const float* X, int n_nodes, int Kh, int Pd, const int* pos, int vocab, const float* Wp, int Fo, float drop_p,
                      const unsigned* mask, const float* d_hw, int need_dh, int act_on, float act_slope, float* d_X, float* dW,
                      float* dP, int x_dropped, void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || Fo < 1 || !X || !Wp || !d_hw || !dW || !ws) return TXE_ERR_ARG;
    if ((need_dh || Pd > 0) && !d_X) return TXE_ERR_ARG;
    if (Pd > 0 && (!pos || !dP || vocab < 1 || vocab > MAX_VOCAB)) return TXE_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return TXE_ERR_ARG;
    const int Kt = Kh + Pd, Kp = round_up(Kt, 32), Fop = round_up(Fo, 32);
    const int St = gcn_dwt_splits(n_nodes, Kp, Fop);
    DenseWs p = plan_dense_ws(ws, n_nodes, Kp, 0, Fop, Pd, vocab, false, St);   // part: [S][Kp][Fop] (or transposed: [St][Fop][Kp])
    if (ws_bytes < p.total) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const int c0 = need_dh ? 0 : (Kh / 4) * 4;
    if (Kt - c0 > 0 && n_nodes > 0 && (need_dh || Pd > 0)) {
        // d_X[m][c] = sum_f d_hw[m][f] * Wp[c][f]     (NT; Wp rows are padded to a multiple of 128, so every tile is plain)
        VMat A = vmat_plain(d_hw, Fop, n_nodes, Fop);
        VMat B = vmat_plain(Wp + (long long)c0 * Fop, Fop, round_up(Kp, 128) - c0, Fop);
        Epi E = epi_plain(d_X + c0, Kp, Kh > c0 ? Kh - c0 : 0);
        E.c2 = d_X + c0 + E.cols_main; E.ldc2 = Kp;
        epi_set_mask(E, mask, Kt, c0, drop_p);
        if (act_on && need_dh) epi_set_act(E, X + c0, Kp, act_slope);
        E.alg_flops = 2.0 * n_nodes * (double)(need_dh ? Kt : Pd) * Fo;
        rc = gemm_nt(A, B, E, n_nodes, Kt - c0, Fop, 1, s, p.tail, p.tail_bytes);
        if (rc) return rc;
    }
    if (Pd > 0) {
        if (n_nodes > 0) {
            hipLaunchKernelGGL(pos_segsum_stage1, dim3(p.seg_blocks), dim3(256), 0, s, (const float*)(d_X + Kh), (long long)Kp, pos, n_nodes,
                               Pd, vocab, p.seg_rows, p.ppart);
            TXE_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(pos_segsum_stage2, dim3((vocab * Pd + 63) / 64), dim3(256), 0, s, (const float*)p.ppart,
                           n_nodes > 0 ? p.seg_blocks : 0, vocab, Pd, dP);
        TXE_CHECK_LAUNCH();
    }
    if (St > 0 && (x_dropped || !mask || drop_p <= 0.f)) {
        // dW^T [Fop][Kp] = d_hw^T X on 128 x 160 tiles (gcn_dwt_splits), written back transposed by the slice reduction
        VMat A = vmat_plain(d_hw, Fop, n_nodes, Fop);
        VMat B = vmat_plain(X, Kp, n_nodes, Kp);
        Epi E = epi_plain(p.part, Kp, Kp);
        E.split_stride = (long long)Fop * Kp;
        E.alg_flops = 2.0 * Kt * (double)Fo * n_nodes;
        E.route |= GEMM_ROUTE_FORCE_BN160;
        rc = gemm_tn(A, B, E, Fop, Kp, n_nodes, St, s);
        if (rc) return rc;
        hipLaunchKernelGGL(reduce_splits_transposed_kernel, dim3((Fo + 7) / 8, (Kt + 31) / 32), dim3(256), 0, s, (const float*)p.part, St,
                           E.split_stride, Kt, Fo, Kp, dW);
        TXE_CHECK_LAUNCH();
    } else {   // dWp[k][f] = sum_m dropout(X)[m][k] * d_hw[m][f]
        VMat A = vmat_plain(X, Kp, n_nodes, Kp);
        if (!x_dropped) vmat_set_mask(A, mask, drop_p);             // (x_dropped: X already holds dropout(X), txe_gcn_layer_prepare)
        VMat B = vmat_plain(d_hw, Fop, n_nodes, Fop);
        Epi E = epi_plain(p.part, Fop, Fop);
        E.split_stride = (long long)Kp * Fop;
        E.alg_flops = 2.0 * Kt * (double)Fo * n_nodes;
        rc = gemm_tn(A, B, E, Kp, Fop, n_nodes, p.splits, s);
        if (rc) return rc;
        const long long n = (long long)Kt * Fo;
        hipLaunchKernelGGL(reduce_splits_sub_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s,
                           (const float*)p.part, n_nodes > 0 ? p.splits : 0, E.split_stride, Kt, Fo, Fop, dW);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

}  // extern "C"

// =====================================================================================================================
// Last GATLayer folded behind a linear readout (PGAT / GAT output layer with ONE head + MeanReadout / WeightedMeanReadout;
// model_zoo.py:80-104,219,227-242).  The output layer has no activation, its head mean is the identity, and the readout is
// a weighted mean, so   hg[g] = sum_v w_v/S_g * sum_u alpha'_uv * (Xd[u] W^T)  =  ( sum_{u in g} c_u Xd[u] ) W^T,
//     c_u = sum_{v : u->v} w_v alpha'_uv / S_g,   Xd = feat-dropped layer input,  alpha' = attention-dropped softmax,
// and the attention logits need only two columns:  a1 = Xd wa1, a2 = Xd wa2 (the folded rows F, F+1 of Wp).
// Same arithmetic, different association: the projection (and its dX / dW products) shrink from N node rows to G graph
// rows (4.4x fewer flops on the MAG batch); everything else is two HBM sweeps over X in forward and two in backward.
//   forward : logits (sweep 1) -> alpha [E] -> c~ [N] -> Z[g] = sum c~_u Xd[u] / S_g (sweep 2) -> hg = Z W^T (GEMM, G rows)
//   backward: dZ = d_hg W, dW = d_hg^T Z (GEMMs, G rows) -> dc~_u = <dZ[g], Xd[u]>/S_g, dS_g (sweep 3) -> edge-level softmax /
//             readout-weight backward -> d_X[u] = keep*s*(c_u dZ[g] + da1_u wa1 + da2_u wa2) * leaky'(X), d_wa (sweep 4) -> unfold.
// =====================================================================================================================
namespace txe {

constexpr int CL_NI = 4;                      // 16-byte vectors per lane per column tile (256 vectors = 1024 columns per tile)

__device__ __forceinline__ float cl_softplus(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float cl_sigmoid(float x) { return x > 20.f ? 1.f : 1.f / (1.f + __expf(-x)); }

// keep factors (0 / 1) of the 4 columns of vector j from the row's mask words (mask == nullptr: all kept)
template <bool MASK>
__device__ __forceinline__ void cl_keep4(const unsigned* __restrict__ mrow, int mask_ld, int j, float* k4) {
    if constexpr (!MASK) { k4[0] = k4[1] = k4[2] = k4[3] = 1.f; return; }
    // vector j < Kp / 4 and the mask row has Kp / 32 = mask_ld words: the word always exists.  (A bounds select here makes hipcc sink
    // the load into the conditional and wait vmcnt(0) behind it -- one load in flight per wave.)
    const int c = j * 4;
    const unsigned b = mrow[c >> 5] >> (c & 31);
    k4[0] = (b & 1u) ? 1.f : 0.f; k4[1] = (b & 2u) ? 1.f : 0.f; k4[2] = (b & 4u) ? 1.f : 0.f; k4[3] = (b & 8u) ? 1.f : 0.f;
}

// sweep 1 -- one wave per node (persistent waves keep the two folded rows in registers per column tile):
//   a12[u][0] = <Xd[u], wa1>,  a12[u][1] = <Xd[u], wa2>
template <bool MASK>
__global__ __launch_bounds__(256) void cl_logits_kernel(const float* __restrict__ X, int Kp, int n_nodes, const unsigned* __restrict__ mask,
                                                        int mask_ld, float scale, const float* __restrict__ wa /*[2][Kp]*/,
                                                        float* __restrict__ a12) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    const int nvec = Kp >> 2;
    for (int u = wave; u < n_nodes; u += nwaves) {
        const float* row = X + (long long)u * Kp;
        const unsigned* mrow = mask + (MASK ? (long long)u * mask_ld : 0);
        float s1 = 0.f, s2 = 0.f;
        for (int t0 = 0; t0 < nvec; t0 += 64 * CL_NI) {
            float x[CL_NI][4], w1[CL_NI][4], w2[CL_NI][4], k4[CL_NI][4];
#pragma unroll
            for (int i = 0; i < CL_NI; ++i) {
                const int j = t0 + l + 64 * i;
                const int jc = (j < nvec) ? j : t0;
                vload<4>(row + jc * 4, x[i]);
                vload<4>(wa + jc * 4, w1[i]);
                vload<4>(wa + Kp + jc * 4, w2[i]);
                cl_keep4<MASK>(mrow, mask_ld, jc, k4[i]);
                const float live = (j < nvec) ? 1.f : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) k4[i][k] *= live;
            }
#pragma unroll
            for (int i = 0; i < CL_NI; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xd = x[i][k] * k4[i][k];
                    s1 = fmaf(xd, w1[i][k], s1);
                    s2 = fmaf(xd, w2[i][k], s2);
                }
        }
        s1 = wave_sum(s1) * scale;
        s2 = wave_sum(s2) * scale;
        if (l == 0) { a12[2 * (long long)u] = s1; a12[2 * (long long)u + 1] = s2; }
    }
}

// per graph: S_g = sum_v w_v -> wsum[g];  gid[v] = g for its nodes
__device__ __forceinline__ void cl_wsum_job(const int bid, const int* __restrict__ goff, int G, const int* __restrict__ pos,
                                            const float* __restrict__ pw, float* __restrict__ wsum, int* __restrict__ gid) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g = bid * 4 + w;
    if (g >= G) return;
    const int beg = goff[g], end = goff[g + 1];
    float S = 0.f;
    for (int v = beg + l; v < end; v += 64) {
        S += pw ? cl_softplus(pw[pos[v]]) : 1.f;
        gid[v] = g;
    }
    S = wave_sum(S);
    if (l == 0) wsum[g] = S;
}
__global__ __launch_bounds__(256) void cl_wsum_kernel(const int* __restrict__ goff, int G, const int* __restrict__ pos,
                                                      const float* __restrict__ pw, float* __restrict__ wsum, int* __restrict__ gid) {
    cl_wsum_job(blockIdx.x, goff, G, pos, pw, wsum, gid);
}
// sweep 2 -- one wave per (graph, 256-column tile):  Z[g][tile] = (scale / S_g) sum_{u in g} c~_u (X[u] * keep)[tile]
template <bool MASK>
__global__ __launch_bounds__(256) void cl_zsum_kernel(const int* __restrict__ goff, int G, int ntile, const float* __restrict__ X, int Kp,
                                                      const unsigned* __restrict__ mask, int mask_ld, float scale,
                                                      const float* __restrict__ coef, const float* __restrict__ wsum,
                                                      float* __restrict__ Z) {
    const int l = threadIdx.x & 63;
    const long long wid = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int g = (int)(wid / ntile), t = (int)(wid % ntile);
    if (g >= G) return;
    const int beg = goff[g], end = goff[g + 1];
    const int nvec = Kp >> 2;
    const int j = t * 64 + l;
    const int jc = (j < nvec) ? j : t * 64;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int u0 = beg; u0 < end; u0 += 4) {                          // four nodes per step: independent loads in flight
        float x[4][4], k4[4][4], cu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = min(u0 + e, end - 1);
            cu[e] = coef[u] * ((u0 + e < end) ? 1.f : 0.f);
            vload<4>(X + (long long)u * Kp + jc * 4, x[e]);
            cl_keep4<MASK>(mask + (MASK ? (long long)u * mask_ld : 0), mask_ld, jc, k4[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(cu[e] * k4[e][k], x[e][k], acc[k]);
    }
    if (j < nvec) {
        const float S = wsum[g];
        const float zs = S > 0.f ? scale / S : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] *= zs;
        vstore<4>(Z + (long long)g * Kp + j * 4, acc);
    }
}

// The same sweep with one wave per (chunk of ZS_GPW consecutive graphs, column tile): an egonet has ~4 nodes, so a (graph, tile) wave
// asks for 4 KB and is gone -- 36,864 waves of two dependent round trips each on the training batch.  A chunk's nodes are one
// contiguous range: the wave streams it eight nodes (8 KB) per step and writes a graph's row of Z whenever the range crosses into
// the next graph (offsets and weight sums of the chunk sit in lanes, read back as scalars: uniform control flow).  Per graph the same
// nodes in the same order: bit-identical to cl_zsum_kernel.
#ifndef TXE_ZS_GPW
#define TXE_ZS_GPW 4
#endif
constexpr int ZS_GPW = TXE_ZS_GPW;
// EDOT (the graph vector folded into a bilinear matcher): the gradient of Z will be dZ[g] = dsl_g Tf[zrow[g]] with Tf known NOW, so
// the backward's <dZ[g], keep X[u]> sweep is this sweep's <Tf[zrow[g]], keep X[u]> times a scalar: the wave adds its tile's share of that
// dot product per node to e_part[u][tile] (summed over the tiles, in tile order, by cl_fold_dc_kernel).
template <bool MASK, bool EDOT = false>
__global__ __launch_bounds__(256) void cl_zsum_chunk_kernel(const int* __restrict__ goff, int G, int ntile, int nmap, const float* __restrict__ X, int Kp,
                                                            const unsigned* __restrict__ mask, int mask_ld, float scale,
                                                            const float* __restrict__ coef, const float* __restrict__ wsum,
                                                            float* __restrict__ Z, const float* __restrict__ Tf = nullptr,
                                                            const int* __restrict__ zrow = nullptr, float* __restrict__ e_part = nullptr) {
    constexpr int NU = 8;
    const int l = threadIdx.x & 63;
    const long long wid = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    // waves are numbered over (chunk, nmap slots): nmap = ntile, or ntile + 1 with an idle slot when ntile is a multiple of 4 -- the four
    // waves of a workgroup (and the two workgroups of an 8-tile row) would otherwise always sit on the SAME chunk's rows, which costs a
    // quarter of the sweep's rate (8 tiles: 67 against 51 us; 4: 35 / 25; 16: 118 / 90 -- with or without the e_part stores)
    const int ch = (int)(wid / nmap), t = (int)(wid % nmap);
    const int g0 = ch * ZS_GPW;
    if (g0 >= G || t >= ntile) return;
    const int ng = min(ZS_GPW, G - g0);
    const int my_off = goff[g0 + min(l, ng)];                       // lanes 0..ng: the chunk's graph offsets
    const float my_ws = wsum[g0 + min(l, ng - 1)];                  // lanes 0..ng-1: their weight sums
    const int nvec = Kp >> 2;
    const int j = t * 64 + l;
    const int jc = (j < nvec) ? j : t * 64;
    const int beg = __builtin_amdgcn_readlane(my_off, 0), end = __builtin_amdgcn_readlane(my_off, ng);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int gi = 0;                                                     // current graph of the chunk (uniform)
    int next = __builtin_amdgcn_readlane(my_off, 1);               // first node past it
    float tt[4] = {0.f, 0.f, 0.f, 0.f};                             // EDOT: this lane's piece of Tf[zrow[current graph]]
    int my_zr = 0;
    if constexpr (EDOT) {
        my_zr = zrow[g0 + min(l, ng - 1)];                          // lanes 0..ng-1: the chunk's rows of Tf
        vload<4>(Tf + (long long)__builtin_amdgcn_readlane(my_zr, 0) * Kp + jc * 4, tt);
    }
    auto flush = [&]() {                                            // graph gi is complete: scale, store, start the next one
        const float S = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_ws), gi));
        const float zs = S > 0.f ? scale / S : 0.f;
        if (j < nvec) {
            float o[4] = {acc[0] * zs, acc[1] * zs, acc[2] * zs, acc[3] * zs};
            vstore<4>(Z + (long long)(g0 + gi) * Kp + j * 4, o);
        }
        acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
        ++gi;
        next = __builtin_amdgcn_readlane(my_off, min(gi + 1, ng));
        if constexpr (EDOT) vload<4>(Tf + (long long)__builtin_amdgcn_readlane(my_zr, min(gi, ng - 1)) * Kp + jc * 4, tt);
    };
    for (int u0 = beg; u0 < end; u0 += NU) {                        // NU nodes per step: independent loads in flight
        float x[NU][4], k4[NU][4], cu[NU];
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int u = min(u0 + e, end - 1);
            cu[e] = coef[u];
            vload<4>(X + (long long)u * Kp + jc * 4, x[e]);
            cl_keep4<MASK>(mask + (MASK ? (long long)u * mask_ld : 0), mask_ld, jc, k4[e]);
        }
        float pe[NU];                                               // EDOT: this lane's share of the NU nodes' dot products with Tf
#pragma unroll
        for (int e = 0; e < NU; ++e) pe[e] = 0.f;
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int u = u0 + e;
            if (u < end) {                                          // (uniform)
                while (u >= next) flush();                          // graphs that ended before u (empty ones included)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(cu[e] * k4[e][k], x[e][k], acc[k]);
                if constexpr (EDOT) {
                    float q = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) q = fmaf(tt[k] * k4[e][k], x[e][k], q);
                    pe[e] = (j < nvec) ? q : 0.f;
                }
            }
        }
        if constexpr (EDOT) {
            // eight sums over the wave in 10 exchanges instead of 8 x 6: halve the set of values a lane carries with every exchange
            // (lane bit 5 picks nodes 0-3 / 4-7, bit 4 pairs, bit 3 one), then three plain butterflies; lane 8 n holds node n's sum
            static_assert(NU == 8, "the reduction below is written for eight nodes per step");
            // (all on the VALU: v_permlane32_swap / v_permlane16_swap hand the half a lane does not keep to its partner 32 / 16 lanes away,
            //  DPP row rotations and quad permutes do the rest -- __shfl_xor is ds_bpermute, a trip through the LDS pipeline per exchange;
            //  same pairs added in the same order)
            float a4[4], b2[2];
            const bool h5 = (l & 32) != 0, h4 = (l & 16) != 0, h3 = (l & 8) != 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(pe[k]), __float_as_uint(pe[k + 4]), false, false);
                a4[k] = h5 ? __uint_as_float(r[1]) + __uint_as_float(r[0]) : __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a4[k]), __float_as_uint(a4[k + 2]), false, false);
                b2[k] = h4 ? __uint_as_float(r[1]) + __uint_as_float(r[0]) : __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            float c1 = (h3 ? b2[1] : b2[0]) + dpp_f<0x128>(h3 ? b2[0] : b2[1]);          // row_ror:8 = lane ^ 8
            {   // lane ^ 4: row_shl:4 for the lanes with bit 2 clear (banks 0, 2), row_shr:4 for the others
                int o = __builtin_amdgcn_update_dpp(0, __float_as_int(c1), 0x104, 0xF, 0x5, false);
                o = __builtin_amdgcn_update_dpp(o, __float_as_int(c1), 0x114, 0xF, 0xA, false);
                c1 += __int_as_float(o);
            }
            c1 += dpp_f<0x4E>(c1);                                                       // quad_perm [2,3,0,1] = lane ^ 2
            c1 += dpp_f<0xB1>(c1);                                                       // quad_perm [1,0,3,2] = lane ^ 1
            const int en = (h5 ? 4 : 0) + (h4 ? 2 : 0) + (h3 ? 1 : 0);
            if ((l & 7) == 0 && u0 + en < end) e_part[(long long)(u0 + en) * ntile + t] = c1;
        }
    }
    while (gi < ng) flush();                                        // the last graph, and empty graphs at the chunk's end
}

// launch of the Z sweep: small graphs (egonets: ~4 nodes) on the chunked kernel, large ones one wave per graph and tile
static bool cl_zsum_chunked(int n_nodes, int G) { return (long long)n_nodes <= 16LL * G && G >= 16; }
static int cl_zsum_launch(const int* graph_off, int G, int n_nodes, const float* X, int Kp, const unsigned* mk, const unsigned* dummy_mask,
                          int mask_ld, float fs, const float* coef, const float* wsum, float* Z, hipStream_t s, const float* Tf = nullptr,
                          const int* zrow = nullptr, float* e_part = nullptr) {
    const int ntile = (Kp / 4 + 63) / 64;
    const int nmap = (ntile % 4 == 0) ? ntile + 1 : ntile;          // (slots per chunk in the chunked kernel's wave numbering: see there)
    const bool chunked = cl_zsum_chunked(n_nodes, G);
    if (e_part) {                                   // (only the chunked kernel forms the dot products: the entry point checks cl_zsum_chunked)
        if (!chunked || !Tf || !zrow) return TXE_ERR_ARG;
        const long long nw = (long long)((G + ZS_GPW - 1) / ZS_GPW) * nmap;
        ProfScope prof(mk ? "cl_zsum_chunk_kernel<true, true>" : "cl_zsum_chunk_kernel<false, true>", s, 4.0 * (n_nodes + (double)G) * Kp, 1);
        const dim3 grid((unsigned)((nw + 3) / 4));
        if (mk) hipLaunchKernelGGL((cl_zsum_chunk_kernel<true, true>), grid, dim3(256), 0, s, graph_off, G, ntile, nmap, X, Kp, mk, mask_ld, fs, coef, wsum, Z, Tf, zrow, e_part);
        else hipLaunchKernelGGL((cl_zsum_chunk_kernel<false, true>), grid, dim3(256), 0, s, graph_off, G, ntile, nmap, X, Kp, dummy_mask, mask_ld, fs, coef, wsum, Z, Tf,
                                zrow, e_part);
        TXE_CHECK_LAUNCH();
        return TXE_OK;
    }
    const long long nwaves = chunked ? (long long)((G + ZS_GPW - 1) / ZS_GPW) * nmap : (long long)G * ntile;
    ProfScope prof(chunked ? (mk ? "cl_zsum_chunk_kernel<true, false>" : "cl_zsum_chunk_kernel<false, false>") : (mk ? "cl_zsum_kernel<true>" : "cl_zsum_kernel<false>"), s,
                   4.0 * (n_nodes + (double)G) * Kp, 1);
    const dim3 grid((unsigned)((nwaves + 3) / 4));
    if (chunked && mk) hipLaunchKernelGGL((cl_zsum_chunk_kernel<true, false>), grid, dim3(256), 0, s, graph_off, G, ntile, nmap, X, Kp, mk, mask_ld, fs, coef, wsum, Z,
                                          (const float*)nullptr, (const int*)nullptr, (float*)nullptr);
    else if (chunked) hipLaunchKernelGGL((cl_zsum_chunk_kernel<false, false>), grid, dim3(256), 0, s, graph_off, G, ntile, nmap, X, Kp, dummy_mask, mask_ld, fs, coef, wsum,
                                         Z, (const float*)nullptr, (const int*)nullptr, (float*)nullptr);
    else if (mk) hipLaunchKernelGGL(cl_zsum_kernel<true>, grid, dim3(256), 0, s, graph_off, G, ntile, X, Kp, mk, mask_ld, fs, coef, wsum, Z);
    else hipLaunchKernelGGL(cl_zsum_kernel<false>, grid, dim3(256), 0, s, graph_off, G, ntile, X, Kp, dummy_mask, mask_ld, fs, coef, wsum, Z);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// per graph: dS[g] = -<dZ[g], Z[g]> / S_g
__global__ __launch_bounds__(256) void cl_bwd_ds_kernel(int G, int Kp, const float* __restrict__ dZ, const float* __restrict__ Z,
                                                        const float* __restrict__ wsum, float* __restrict__ dS) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + w;
    if (g >= G) return;
    const int nvec = Kp >> 2;
    float s = 0.f;
    for (int j = l; j < nvec; j += 64) {
        float d[4], z[4];
        vload<4>(dZ + (long long)g * Kp + j * 4, d);
        vload<4>(Z + (long long)g * Kp + j * 4, z);
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fmaf(d[k], z[k], s);
    }
    s = wave_sum(s);
    if (l == 0) dS[g] = wsum[g] > 0.f ? -s / wsum[g] : 0.f;
}

// sweep 3 -- one wave per node:  dc~_u = (scale / S_g) <dZ[g], X[u] * keep>
template <bool MASK>
__global__ __launch_bounds__(256) void cl_bwd_dot_kernel(int n_nodes, const int* __restrict__ gid, const float* __restrict__ X, int Kp,
                                                         const unsigned* __restrict__ mask, int mask_ld, float scale,
                                                         const float* __restrict__ dZ, const float* __restrict__ wsum,
                                                         const float* __restrict__ coef, float* __restrict__ dc, float* __restrict__ cn,
                                                         const int nb_ds, const int G, const int D, const float* __restrict__ d_hg,
                                                         const long long ld_dhg, const float* __restrict__ hg, const long long ld_hg,
                                                         float* __restrict__ dS) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if ((int)blockIdx.x < nb_ds) {
        // independent job on the first workgroups, one wave per graph: dS[g] = -<dZ[g], Z[g]> / S_g, and since dZ = d_hg W and
        // hg = Z W^T the product is <d_hg[g], hg[g]> -- D columns instead of Kp, and no dependence on the dZ GEMM
        const int g = blockIdx.x * 4 + w;
        if (g >= G) return;
        float s = 0.f;
        for (int j = l; j < D; j += 64) s = fmaf(d_hg[(long long)g * ld_dhg + j], hg[(long long)g * ld_hg + j], s);
        s = wave_sum(s);
        if (l == 0) dS[g] = wsum[g] > 0.f ? -s / wsum[g] : 0.f;
        return;
    }
    const int u = ((int)blockIdx.x - nb_ds) * 4 + w;
    if (u >= n_nodes) return;
    const int g = gid[u];
    const int nvec = Kp >> 2;
    const float* row = X + (long long)u * Kp;
    const float* dzrow = dZ + (long long)g * Kp;
    const unsigned* mrow = mask + (MASK ? (long long)u * mask_ld : 0);
    float part = 0.f;
    for (int t0 = 0; t0 < nvec; t0 += 64 * CL_NI) {
        float x[CL_NI][4], d[CL_NI][4], k4[CL_NI][4];
#pragma unroll
        for (int i = 0; i < CL_NI; ++i) {
            const int j = t0 + l + 64 * i;
            const int jc = (j < nvec) ? j : t0;
            vload<4>(row + jc * 4, x[i]);
            vload<4>(dzrow + jc * 4, d[i]);
            cl_keep4<MASK>(mrow, mask_ld, jc, k4[i]);
            const float live = (j < nvec) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) k4[i][k] *= live;
        }
#pragma unroll
        for (int i = 0; i < CL_NI; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) part = fmaf(d[i][k] * k4[i][k], x[i][k], part);
    }
    part = wave_sum(part);
    if (l == 0) {
        const float S = wsum[g];
        const float inv = S > 0.f ? 1.f / S : 0.f;
        dc[u] = part * scale * inv;
        cn[u] = coef[u] * inv;                        // normalised coefficient for the d_X sweep
    }
}

// The same sweep with the node's WHOLE row in one round trip: NT tiles of 64 vectors per lane issued together (the tile loop above is
// three dependent round trips for a 2,080-column row, behind the gid one).  Lanes past the row re-read its first tile (L1 hits) with a
// zero factor; per lane the vectors are summed in the same ascending order: bit-identical.
template <bool MASK, int NT>
__global__ __launch_bounds__(256) void cl_bwd_dot_row_kernel(int n_nodes, const int* __restrict__ gid, const float* __restrict__ X, int Kp,
                                                             const unsigned* __restrict__ mask, int mask_ld, float scale,
                                                             const float* __restrict__ dZ, const float* __restrict__ wsum,
                                                             const float* __restrict__ coef, float* __restrict__ dc, float* __restrict__ cn,
                                                             const int nb_ds, const int G, const int D, const float* __restrict__ d_hg,
                                                             const long long ld_dhg, const float* __restrict__ hg, const long long ld_hg,
                                                             float* __restrict__ dS) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if ((int)blockIdx.x < nb_ds) {                      // (the dS job of cl_bwd_dot_kernel)
        const int g = blockIdx.x * 4 + w;
        if (g >= G) return;
        float s = 0.f;
        for (int j = l; j < D; j += 64) s = fmaf(d_hg[(long long)g * ld_dhg + j], hg[(long long)g * ld_hg + j], s);
        s = wave_sum(s);
        if (l == 0) dS[g] = wsum[g] > 0.f ? -s / wsum[g] : 0.f;
        return;
    }
    const int u = ((int)blockIdx.x - nb_ds) * 4 + w;
    if (u >= n_nodes) return;
    const int g = gid[u];
    const float S = wsum[g];
    const float cu = coef[u];
    const int nvec = Kp >> 2;                           // (> 64: the launcher sends narrower rows to cl_bwd_dot_kernel)
    const float* row = X + (long long)u * Kp;
    const float* dzrow = dZ + (long long)g * Kp;
    const unsigned* mrow = mask + (MASK ? (long long)u * mask_ld : 0);
    float x[NT][4], d[NT][4], k4[NT][4];
    // (the node's own row first and the gid-dependent dZ row behind it, in two loops: 61 against 55 us)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int j = l + 64 * i;
        const int jc = (j < nvec) ? j : l;
        vload<4>(row + jc * 4, x[i]);
        vload<4>(dzrow + jc * 4, d[i]);
        cl_keep4<MASK>(mrow, mask_ld, jc, k4[i]);
        const float live = (j < nvec) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) k4[i][k] *= live;
    }
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) part = fmaf(d[i][k] * k4[i][k], x[i][k], part);
    part = wave_sum(part);
    if (l == 0) {
        const float inv = S > 0.f ? 1.f / S : 0.f;
        dc[u] = part * scale * inv;
        cn[u] = cu * inv;
    }
}

// launch of sweep 3: rows of 65..640 vectors go out in one round trip of 5, 9 or 10 tiles (cl_bwd_dot_row_kernel), anything else tile by tile
static int cl_bwd_dot_launch(int n_nodes, const int* gid, const float* X, int Kp, const unsigned* mk, const unsigned* dummy_mask, int mask_ld, float fs,
                             const float* dZ, const float* wsum, const float* coef, float* dc, float* cn, int nb_ds, int G, int D, const float* d_hg,
                             long long ld_dhg, const float* hg, long long ld_hg, float* dS, double bytes, hipStream_t s) {
    const int nb = (n_nodes + 3) / 4, nvec = Kp >> 2;
    const int nt = (nvec > 64 && nvec <= 320) ? 5 : ((nvec > 320 && nvec <= 576) ? 9 : ((nvec > 576 && nvec <= 640) ? 10 : 0));
    const unsigned* m = mk ? mk : dummy_mask;
    const dim3 grid(nb_ds + nb);
#define TXE_BD_ARGS n_nodes, gid, X, Kp, m, mask_ld, fs, dZ, wsum, coef, dc, cn, nb_ds, G, D, d_hg, ld_dhg, hg, ld_hg, dS
    if (nt == 0) {
        ProfScope prof(mk ? "cl_bwd_dot_kernel<true>" : "cl_bwd_dot_kernel<false>", s, bytes, 1);
        if (mk) hipLaunchKernelGGL(cl_bwd_dot_kernel<true>, grid, dim3(256), 0, s, TXE_BD_ARGS);
        else hipLaunchKernelGGL(cl_bwd_dot_kernel<false>, grid, dim3(256), 0, s, TXE_BD_ARGS);
    } else {
        static const char* names[6] = {"cl_bwd_dot_row_kernel<false, 5>", "cl_bwd_dot_row_kernel<true, 5>", "cl_bwd_dot_row_kernel<false, 9>",
                                       "cl_bwd_dot_row_kernel<true, 9>", "cl_bwd_dot_row_kernel<false, 10>", "cl_bwd_dot_row_kernel<true, 10>"};
        ProfScope prof(names[(mk ? 1 : 0) + (nt == 9 ? 2 : (nt == 10 ? 4 : 0))], s, bytes, 1);
        if (nt == 5 && mk) hipLaunchKernelGGL((cl_bwd_dot_row_kernel<true, 5>), grid, dim3(256), 0, s, TXE_BD_ARGS);
        else if (nt == 5) hipLaunchKernelGGL((cl_bwd_dot_row_kernel<false, 5>), grid, dim3(256), 0, s, TXE_BD_ARGS);
        else if (nt == 9 && mk) hipLaunchKernelGGL((cl_bwd_dot_row_kernel<true, 9>), grid, dim3(256), 0, s, TXE_BD_ARGS);
        else if (nt == 9) hipLaunchKernelGGL((cl_bwd_dot_row_kernel<false, 9>), grid, dim3(256), 0, s, TXE_BD_ARGS);
        else if (mk) hipLaunchKernelGGL((cl_bwd_dot_row_kernel<true, 10>), grid, dim3(256), 0, s, TXE_BD_ARGS);
        else hipLaunchKernelGGL((cl_bwd_dot_row_kernel<false, 10>), grid, dim3(256), 0, s, TXE_BD_ARGS);
    }
#undef TXE_BD_ARGS
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The folded layer's edge-level work as ONE launch each way.  Everything here is tiny (a few bytes per edge) and stays inside a graph,
// so a workgroup that owns CG_GRAPHS whole graphs can run the destination-side and the source-side halves back to back behind a
// workgroup barrier (they were two ~10 us launches each).  Degrees up to CG_LIGHT are walked by one thread per node with every load
// unrolled and clamped (no branch between a load and its use); heavier nodes (an egonet's anchor feeds up to 50 siblings; hubs of
// generic graphs) are collected and handled by a whole wave each.
//   forward : alpha[p] = softmax_in(leaky(a1[u] + a2[v])), gid, w_v;  S_g = sum w_v;  c~_u = sum_out w_v f alpha
//   backward: dz[p], da2[v], dwv[v] (destination side, cl_bwd_edge_kernel's math);  da1[u] = sum_out dz (source side)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int CG_GRAPHS = 8;
constexpr int CG_LIGHT = 8;
constexpr int CG_MAXN = 512;        // nodes of a workgroup whose readout weights are staged in LDS (beyond: recomputed)

__device__ __forceinline__ int cg_graph_of(const int* s_goff, int ng, int v) {
    int g = 0;
#pragma unroll
    for (int q = 1; q < CG_GRAPHS; ++q) g += (q < ng && v >= s_goff[q]) ? 1 : 0;
    return g;
}

__global__ __launch_bounds__(256) void cl_attn_coef_kernel(const int* __restrict__ rowptr_in, const int* __restrict__ col_src,
                                                           const int* __restrict__ rowptr_out, const int* __restrict__ col_dst,
                                                           const int* __restrict__ pos_out, const int* __restrict__ goff, const int G,
                                                           const float* __restrict__ a12, const float slope, const float drop_p,
                                                           const float drop_scale, const unsigned long long seed,
                                                           const int* __restrict__ pos, const float* __restrict__ pw,
                                                           float* __restrict__ alpha, float* __restrict__ coef, float* __restrict__ wsum,
                                                           int* __restrict__ gid) {
    __shared__ int s_goff[CG_GRAPHS + 1], s_heavy[2][256], s_nh[2];
    __shared__ float s_wv[CG_MAXN];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g0 = blockIdx.x * CG_GRAPHS, g1 = min(G, g0 + CG_GRAPHS), ng = g1 - g0;
    if (threadIdx.x <= ng) s_goff[threadIdx.x] = goff[g0 + threadIdx.x];
    if (threadIdx.x < 2) s_nh[threadIdx.x] = 0;
    __syncthreads();
    const int n0 = s_goff[0], nn = s_goff[ng] - n0;
    // ---- destination side: alpha, graph ids, readout weights ----
    for (int t = threadIdx.x; t < nn; t += 256) {
        const int v = n0 + t;
        gid[v] = g0 + cg_graph_of(s_goff, ng, v);
        if (t < CG_MAXN) s_wv[t] = pw ? cl_softplus(pw[pos[v]]) : 1.f;
        const int beg = rowptr_in[v], end = rowptr_in[v + 1];
        if (end - beg > CG_LIGHT) { const int k = atomicAdd(&s_nh[0], 1); if (k < 256) s_heavy[0][k] = v; continue; }
        const float a2v = a12[2 * (long long)v + 1];
        float z[CG_LIGHT];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < CG_LIGHT; ++i) {
            const int p = min(beg + i, max(end - 1, beg));
            const float zz = leaky(a12[2 * (long long)col_src[p]] + a2v, slope);
            z[i] = (beg + i < end) ? zz : -INFINITY;
            m = fmaxf(m, z[i]);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < CG_LIGHT; ++i) { z[i] = (beg + i < end) ? __expf(z[i] - m) : 0.f; sum += z[i]; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < CG_LIGHT; ++i)
            if (beg + i < end) alpha[beg + i] = z[i] * inv;
    }
    __syncthreads();
    {
        const bool listed = s_nh[0] <= 256;
        for (int i = w; i < (listed ? s_nh[0] : nn); i += 4) {
            const int v = listed ? s_heavy[0][i] : n0 + i;
            const int beg = rowptr_in[v], end = rowptr_in[v + 1];
            if (end - beg <= CG_LIGHT) continue;
            const float a2v = a12[2 * (long long)v + 1];
            float m = -INFINITY;
            for (int p = beg + l; p < end; p += 64) m = fmaxf(m, leaky(a12[2 * (long long)col_src[p]] + a2v, slope));
            m = wave_max(m);
            float sum = 0.f;
            for (int p = beg + l; p < end; p += 64) sum += __expf(leaky(a12[2 * (long long)col_src[p]] + a2v, slope) - m);
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
            for (int p = beg + l; p < end; p += 64) alpha[p] = __expf(leaky(a12[2 * (long long)col_src[p]] + a2v, slope) - m) * inv;
        }
    }
    if (threadIdx.x < ng) {                            // S_g: a serial walk in fixed order (deterministic)
        float S = 0.f;
        for (int v = s_goff[threadIdx.x]; v < s_goff[threadIdx.x + 1]; ++v)
            S += (v - n0 < CG_MAXN) ? s_wv[v - n0] : (pw ? cl_softplus(pw[pos[v]]) : 1.f);
        wsum[g0 + threadIdx.x] = S;
    }
    __syncthreads();                                   // alpha of these graphs' edges is complete (first touched below)
    // ---- source side: coefficients ----
    for (int t = threadIdx.x; t < nn; t += 256) {
        const int u = n0 + t;
        const int beg = rowptr_out[u], end = rowptr_out[u + 1];
        if (end - beg > CG_LIGHT) { const int k = atomicAdd(&s_nh[1], 1); if (k < 256) s_heavy[1][k] = u; continue; }
        float cu = 0.f;
#pragma unroll
        for (int i = 0; i < CG_LIGHT; ++i) {
            const int j = min(beg + i, max(end - 1, beg));
            const int p = pos_out[j], v = col_dst[j];
            const int tv = min(max(v - n0, 0), CG_MAXN - 1);
            const float wv = (v - n0 < CG_MAXN && v >= n0) ? s_wv[tv] : (pw ? cl_softplus(pw[pos[v]]) : 1.f);
            const float f = (drop_p > 0.f) ? drop_factor(seed, (unsigned long long)p, drop_p, drop_scale) : 1.f;
            cu += (beg + i < end) ? wv * f * alpha[p] : 0.f;
        }
        coef[u] = cu;
    }
    __syncthreads();
    {
        const bool listed = s_nh[1] <= 256;
        for (int i = w; i < (listed ? s_nh[1] : nn); i += 4) {
            const int u = listed ? s_heavy[1][i] : n0 + i;
            const int beg = rowptr_out[u], end = rowptr_out[u + 1];
            if (end - beg <= CG_LIGHT) continue;
            float cu = 0.f;
            for (int j = beg + l; j < end; j += 64) {
                const int p = pos_out[j], v = col_dst[j];
                const float wv = pw ? cl_softplus(pw[pos[v]]) : 1.f;
                const float f = (drop_p > 0.f) ? drop_factor(seed, (unsigned long long)p, drop_p, drop_scale) : 1.f;
                cu = fmaf(wv * f, alpha[p], cu);
            }
            cu = wave_sum(cu);
            if (l == 0) coef[u] = cu;
        }
    }
}

// The folded matcher's backward in place of the <dZ, X> sweep: dZ[g] = dsl_g Tf[zrow[g]], so
//   dc~_u = dsl_g (scale / S_g) sum_tiles e_part[u][tile],   cn_u = dsl_g c~_u / S_g (the sweep's dZ row is Tf's),   dS_g = -dsl_g raw_g / S_g
// with dsl = ds (* s for the exp matcher) and raw_g = <Z_g, Tf[zrow[g]]> = the score before exp -- per node / per graph scalars of the
// graphs a workgroup of cl_attn_bwd_kernel<true> owns, formed in its prologue (they were a launch of their own, cl_fold_dc_kernel).
struct FoldDcArgs {
    const float* e_part; int ntile; const float *m_ds, *m_s; int m_exp; float scale; const float *wsum, *coef; float *dc, *cn, *dS;
    const int* zrow; int* zgid;
};

template <bool FOLD>
__global__ __launch_bounds__(256) void cl_attn_bwd_kernel(const int* __restrict__ rowptr_in, const int* __restrict__ col_src,
                                                          const int* __restrict__ rowptr_out, const int* __restrict__ pos_out,
                                                          const int* __restrict__ goff, const int G, const float* __restrict__ a12,
                                                          const float slope, const float* __restrict__ alpha, const float drop_p,
                                                          const float drop_scale, const unsigned long long seed,
                                                          const int* __restrict__ pos, const float* __restrict__ pw,
                                                          const float* __restrict__ dc, const float* __restrict__ dS,
                                                          float* __restrict__ dz, float* __restrict__ da1, float* __restrict__ da2,
                                                          float* __restrict__ dwv, const FoldDcArgs fd_) {
    __shared__ int s_goff[CG_GRAPHS + 1], s_heavy[2][256], s_nh[2];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g0 = blockIdx.x * CG_GRAPHS, g1 = min(G, g0 + CG_GRAPHS), ng = g1 - g0;
    if (threadIdx.x <= ng) s_goff[threadIdx.x] = goff[g0 + threadIdx.x];
    if (threadIdx.x < 2) s_nh[threadIdx.x] = 0;
    __syncthreads();
    const int n0 = s_goff[0], nn = s_goff[ng] - n0;
    // (FOLD: dc / dS are written by this workgroup's prologue -- read them back through the same, unrestricted pointers)
    const float* dcp = FOLD ? (const float*)fd_.dc : dc;
    const float* dSp = FOLD ? (const float*)fd_.dS : dS;
    if constexpr (FOLD) {
        if ((int)threadIdx.x < ng) {
            const int g = g0 + threadIdx.x;
            const float sv = fd_.m_s[g], dsl = fd_.m_exp ? fd_.m_ds[g] * sv : fd_.m_ds[g];
            const float raw = fd_.m_exp ? logf(sv) : sv;
            const float S = fd_.wsum[g];
            fd_.dS[g] = (S > 0.f && dsl != 0.f) ? -dsl * raw / S : 0.f;
        }
        for (int t = threadIdx.x; t < nn; t += 256) {
            const int u = n0 + t;
            const int g = g0 + cg_graph_of(s_goff, ng, u);
            const float dsl = fd_.m_exp ? fd_.m_ds[g] * fd_.m_s[g] : fd_.m_ds[g];
            const float S = fd_.wsum[g];
            const float inv = S > 0.f ? 1.f / S : 0.f;
            float e = 0.f;
            for (int q = 0; q < fd_.ntile; ++q) e += fd_.e_part[(long long)u * fd_.ntile + q];
            fd_.dc[u] = dsl * e * fd_.scale * inv;
            // the fused sweep reads "dZ[g]" as Tf[zrow[g]] with dsl_g folded into the node's coefficient: dZ itself is never formed
            fd_.cn[u] = fd_.coef[u] * inv * dsl;
            fd_.zgid[u] = fd_.zrow[g];
        }
        __syncthreads();                               // dc / dS of these graphs: read below by other threads of this workgroup
    }
    // ---- destination side ----
    for (int t = threadIdx.x; t < nn; t += 256) {
        const int v = n0 + t;
        const int beg = rowptr_in[v], end = rowptr_in[v + 1];
        if (end - beg > CG_LIGHT) { const int k = atomicAdd(&s_nh[0], 1); if (k < 256) s_heavy[0][k] = v; continue; }
        const float pwv = pw ? pw[pos[v]] : 0.f;
        const float wv = pw ? cl_softplus(pwv) : 1.f;
        const float a2v = a12[2 * (long long)v + 1];
        float al[CG_LIGHT], fd[CG_LIGHT], zs[CG_LIGHT];
        float T = 0.f, dw = 0.f;
#pragma unroll
        for (int i = 0; i < CG_LIGHT; ++i) {
            const int p = min(beg + i, max(end - 1, beg));
            const int u = col_src[p];
            const float f = (drop_p > 0.f) ? drop_factor(seed, (unsigned long long)p, drop_p, drop_scale) : 1.f;
            const bool ok = beg + i < end;
            al[i] = ok ? alpha[p] : 0.f;
            fd[i] = f * dcp[u];                               // f dc~_u
            zs[i] = a12[2 * (long long)u] + a2v;
            dw += al[i] * fd[i];
            T = fmaf(al[i], wv * fd[i], T);
        }
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CG_LIGHT; ++i) {
            const float gz = al[i] * (wv * fd[i] - T) * (zs[i] > 0.f ? 1.f : slope);
            if (beg + i < end) dz[beg + i] = gz;
            s2 += (beg + i < end) ? gz : 0.f;
        }
        da2[v] = s2;
        dwv[v] = pw ? (dSp[g0 + cg_graph_of(s_goff, ng, v)] + dw) * cl_sigmoid(pwv) : 0.f;
    }
    __syncthreads();
    {
        const bool listed = s_nh[0] <= 256;
        for (int i = w; i < (listed ? s_nh[0] : nn); i += 4) {
            const int v = listed ? s_heavy[0][i] : n0 + i;
            const int beg = rowptr_in[v], end = rowptr_in[v + 1];
            if (end - beg <= CG_LIGHT) continue;
            const float pwv = pw ? pw[pos[v]] : 0.f;
            const float wv = pw ? cl_softplus(pwv) : 1.f;
            const float a2v = a12[2 * (long long)v + 1];
            float T = 0.f, dw = 0.f;
            for (int p = beg + l; p < end; p += 64) {
                const float f = (drop_p > 0.f) ? drop_factor(seed, (unsigned long long)p, drop_p, drop_scale) : 1.f;
                const float gq = alpha[p] * f * dcp[col_src[p]];
                dw += gq;
                T = fmaf(alpha[p], wv * f * dcp[col_src[p]], T);
            }
            T = wave_sum(T);
            dw = wave_sum(dw);
            float s2 = 0.f;
            for (int p = beg + l; p < end; p += 64) {
                const float f = (drop_p > 0.f) ? drop_factor(seed, (unsigned long long)p, drop_p, drop_scale) : 1.f;
                const float de = alpha[p] * (wv * f * dcp[col_src[p]] - T);
                const float zq = a12[2 * (long long)col_src[p]] + a2v;
                const float gz = de * (zq > 0.f ? 1.f : slope);
                dz[p] = gz;
                s2 += gz;
            }
            s2 = wave_sum(s2);
            if (l == 0) {
                da2[v] = s2;
                dwv[v] = pw ? (dSp[g0 + cg_graph_of(s_goff, ng, v)] + dw) * cl_sigmoid(pwv) : 0.f;
            }
        }
    }
    __syncthreads();                                   // dz of these graphs' edges is complete (first touched below)
    // ---- source side ----
    for (int t = threadIdx.x; t < nn; t += 256) {
        const int u = n0 + t;
        const int beg = rowptr_out[u], end = rowptr_out[u + 1];
        if (end - beg > CG_LIGHT) { const int k = atomicAdd(&s_nh[1], 1); if (k < 256) s_heavy[1][k] = u; continue; }
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < CG_LIGHT; ++i) {
            const int j = min(beg + i, max(end - 1, beg));
            a += (beg + i < end) ? dz[pos_out[j]] : 0.f;
        }
        da1[u] = a;
    }
    __syncthreads();
    {
        const bool listed = s_nh[1] <= 256;
        for (int i = w; i < (listed ? s_nh[1] : nn); i += 4) {
            const int u = listed ? s_heavy[1][i] : n0 + i;
            const int beg = rowptr_out[u], end = rowptr_out[u + 1];
            if (end - beg <= CG_LIGHT) continue;
            float a = 0.f;
            for (int j = beg + l; j < end; j += 64) a += dz[pos_out[j]];
            a = wave_sum(a);
            if (l == 0) da1[u] = a;
        }
    }
}

// sweep 4 -- one wave per (chunk of CL_CHUNK nodes, 256-column tile):
//   d_X[u][j] = keep * scale * (c~_u / S_g * dZ[g][j] + da1[u] wa1[j] + da2[u] wa2[j]) * (act_on && j < Kh ? leaky'(X[u][j]) : 1)
//   dwa_part[chunk][0/1][j] = sum over the chunk's nodes of da1/da2[u] * scale * keep * X[u][j]     (fixed order: deterministic)
constexpr int CL_CHUNK = 32;
template <bool MASK, bool ATT>
__global__ __launch_bounds__(256) void cl_bwd_dx_kernel(int n_nodes, int ntile, const int* __restrict__ gid, const float* __restrict__ X, int Kp,
                                                        int Kh, const unsigned* __restrict__ mask, int mask_ld, float scale,
                                                        const float* __restrict__ dZ, const float* __restrict__ cn,
                                                        const float* __restrict__ da1, const float* __restrict__ da2,
                                                        const float* __restrict__ wa, int act_on, float act_slope,
                                                        float* __restrict__ d_X, float* __restrict__ dwa_part) {
    const int l = threadIdx.x & 63;
    const long long wid = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int chunk = (int)(wid / ntile), t = (int)(wid % ntile);
    const int u_beg = chunk * CL_CHUNK;
    if (u_beg >= n_nodes) return;
    const int u_end = min(n_nodes, u_beg + CL_CHUNK);
    const int nvec = Kp >> 2;
    const int j = t * 64 + l;
    const bool jok = j < nvec;
    const int jc = jok ? j : t * 64;
    // leaky' applies to the first Kh columns (the previous layer's activated output); slope 1 elsewhere / when off
    float sl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sl[k] = (act_on && (jc * 4 + k) < Kh) ? act_slope : 1.f;
    float w1[4] = {0.f, 0.f, 0.f, 0.f}, w2[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (ATT) {
        vload<4>(wa + jc * 4, w1);
        vload<4>(wa + Kp + jc * 4, w2);
    }
    constexpr int NU = 8;                            // nodes per step: all their loads are unconditional and issued together
    for (int u0 = u_beg; u0 < u_end; u0 += NU) {
        float x[NU][4], d[NU][4], k4[NU][4], cu[NU], g1[NU], g2[NU];
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int u = min(u0 + e, u_end - 1);
            const float ok = (u0 + e < u_end) ? 1.f : 0.f;
            const int g = gid[u];
            cu[e] = cn[u] * ok;
            g1[e] = ATT ? da1[u] * ok : 0.f;
            g2[e] = ATT ? da2[u] * ok : 0.f;
            vload<4>(X + (long long)u * Kp + jc * 4, x[e]);
            vload<4>(dZ + (long long)g * Kp + jc * 4, d[e]);
            cl_keep4<MASK>(mask + (MASK ? (long long)u * mask_ld : 0), mask_ld, jc, k4[e]);
        }
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ks = k4[e][k] * scale;
                o[k] = ks * (cu[e] * d[e][k] + g1[e] * w1[k] + g2[e] * w2[k]) * ((x[e][k] > 0.f) ? 1.f : sl[k]);
                const float xd = x[e][k] * ks;
                s1[k] = fmaf(g1[e], xd, s1[k]);
                s2[k] = fmaf(g2[e], xd, s2[k]);
            }
            if (jok && u0 + e < u_end) vstore<4>(d_X + (long long)(u0 + e) * Kp + j * 4, o);
        }
    }
    if (ATT && jok) {
        vstore<4>(dwa_part + ((long long)chunk * 2 + 0) * Kp + j * 4, s1);
        vstore<4>(dwa_part + ((long long)chunk * 2 + 1) * Kp + j * 4, s2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Folded layer's d_X sweep FUSED with the previous GATLayer's message/reduce backward (one row sweep instead of three).
// The unfused chain writes d_X' = d(folded layer's input) [N][Kp] (cl_bwd_dx), then reads it twice more: gat_bwd_edge
// (d alpha_e = <d_pre[v], ft[u]>) and gat_bwd_node (d_ft[u] = sum alpha'_e d_pre[v]) -- ~920 MB of HBM traffic on the 18 k-node
// training batch.  But a row of d_pre is an ELEMENTWISE function of rows that are read anyway:
//     d_pre[v][j] = keep[v][j] s (cn_v dZ[g(v)][j] + da1_v wa1[j] + da2_v wa2[j]) leaky'(X'[v][j])          (j < H*D)
// so the source-side sweep can form it on the fly: for source node u, with ft[u] in registers, every out-edge (u -> v) loads X'[v]
// (the row cl_bwd_dx would have read), forms d_pre[v], and uses it twice -- the dot product with ft[u] (d alpha_e) and the
// alpha'-weighted accumulation (d_ft[u]).  d_X' never exists; X', dZ and Y are each read once (+ L2 hits for shared rows), d_Y is
// written once: ~475 MB.  The four waves of a workgroup own a quarter of the H*D row each (for H = 4: one head per wave, so the
// per-head dot products are wave-local); a workgroup walks FB_NODES consecutive source nodes, whose out-edge scalars
// (destination, CSR position, cn, da1, da2) are staged in LDS once, so that the row loads depend on nothing but LDS.
// The per-node leftovers of cl_bwd_dx ride along: the folded attention rows' gradient partials (sum_u da_u Xd'[u]) per workgroup,
// and the position-embedding gradient partials from the (never stored) position columns of d_X'.
// What is left per edge -- softmax / leaky-relu backward of the previous layer's attention from the raw d alpha -- is
// gat_attn_bwd_job (edge-level, a few microseconds; launched together with stage 1 of the reductions).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FB_NODES = 32;        // most source nodes a workgroup walks (fb_nodes_per_wg picks the number for a batch)
constexpr int FB_MAXE = 192;        // out-edges of a workgroup whose scalars are staged in LDS (beyond: read from global)
#ifndef TXE_FB_EU
#define TXE_FB_EU 4
#endif
#ifndef TXE_FB_OCC
#define TXE_FB_OCC 3
#endif
constexpr int FB_EU = TXE_FB_EU;      // out-edges per round trip behind a node's first two
constexpr int FB_MAXPD = 128;       // position columns (Kp - Kh <= 128 is a precondition of the fused path)

struct FusedBwdArgs {
    const int *rowptr_out, *col_dst, *pos_out, *gid, *pos;
    int n_nodes;
    const float* X; int Kp, Kh, Pd; const unsigned* mask; int mask_ld; float fscale;
    const float *dZ, *cn, *da1, *da2, *wa; float act_slope; int vocab;
    const float* Y; long long ld_y; int H, D; const float* alpha; float drop_p, drop_scale; unsigned long long seed;
    float* d_Y; long long ld_dy; float* dal; float* dwa_part; float* ppart;
    int npw;                            // source nodes per workgroup
    // (the egonet-walking variant) the graphs themselves: destination CSR, graph offsets, node -> graph
    const int *rowptr_in, *col_src, *goff, *ggid; int G;
    float* hpart;                       // [workgroups][H*D]: a workgroup's share of d_ft[hub] for a graph whose hub lives in an earlier window
    const int* plan;                    // [n_nodes][8] or NULL: the batch's walk plan (egonet_walk_plan_kernel): the shape checks done once
};

// Source nodes per workgroup of the fused sweep.  The kernel holds 3 workgroups per CU; its workgroups cost about (nodes + 6) each (LDS
// staging of the folded rows, the partial rows written at the end), and a last partial round costs a whole one: on the 18 k-node
// training batch 24 nodes make 745 workgroups = one round of 768 (141 us), 16 make 1.46 rounds (153 us), 32 three quarters of one (154 us).
static inline int fb_nodes_per_wg(int n_nodes, int occupancy = 3) {
    const int slots = occupancy * device_cu_count();
    int best = 16;
    double best_cost = 1e30;
    for (int npw = 12; npw <= FB_NODES; npw += 2) {
        const long long blocks = ((long long)n_nodes + npw - 1) / npw;
        const double cost = (double)((blocks + slots - 1) / slots) * (npw + 6.0);
        if (cost <= best_cost) { best_cost = cost; best = npw; }     // (ties: fewer, larger workgroups)
    }
    return best;
}

// keep bits (low 4) of the 4 columns starting at c (multiple of 4) of row r; all ones without a mask
template <bool MASK>
__device__ __forceinline__ unsigned fb_keep(const unsigned* __restrict__ mask, int mask_ld, long long r, int c) {
    if constexpr (!MASK) return 0xFu;
    // c < Kp = 32 * mask_ld always (the mask has one word per 32 columns of the PADDED row), so the word exists: no bounds select
    // here -- with one, hipcc sinks the load into the conditional and waits vmcnt(0) right behind it, serialising every row load
    return (mask[r * mask_ld + (c >> 5)] >> (c & 31)) & 0xFu;
}

// One workgroup's share of the fused sweep.  STAGED: the out-edge scalars of its FB_NODES source nodes sit in LDS (the usual case);
// otherwise (more than FB_MAXE out-edges) they are read from global memory with dependent loads -- correct, slow, rare.
// There is NO branch between a load and its first use (hipcc waits vmcnt(0) at every control-flow merge behind a pending load):
// every address is clamped to something readable, conditions become weights of 0.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

template <int NI, int EU> struct FbGroup {
    int p[EU];
    float cnv[EU], g1v[EU], g2v[EU], al[EU];
    float xv[EU][NI][4];
    unsigned mv[EU][NI];
};

template <bool MASK, int NI, int NWH, bool STAGED>
__device__ __forceinline__ void fb_body(const FusedBwdArgs& a, const int b, const int u0, const int u1, const int e0, const int ne,
                                        const int* s_v, const int* s_p, const float* s_cn, const float* s_g1, const float* s_g2,
                                        const int* s_ni, const float* s_nf, float (*s_dot)[4], float* s_dp, const float* s_wa,
                                        float* s_acc) {
    const int w = uni((int)(threadIdx.x >> 6)), l = threadIdx.x & 63;   // w is wave-uniform: say so (SGPRs, scalar ALU)
    const int F = a.H * a.D, SL = F >> 2, nvec = SL >> 2;
    const int c0 = w * SL, hw = c0 / a.D;
    const int Kp = a.Kp;
    int off[NI];
    float live[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = l + 64 * i;
        off[i] = c0 + 4 * ((j < nvec) ? j : 0);
        live[i] = (j < nvec) ? 1.f : 0.f;
    }
    // tail columns [F, Kp) (position embedding + padding) of the folded layer's input: lanes of wave 0 (the loads are issued by
    // every lane with clamped addresses; only the tail lanes use them)
    const int tvec = (Kp - F) >> 2;
    const bool tail = (w == 0) && (l < tvec);
    const int tc = min(F + 4 * (tail ? l : 0), Kp - 4);
    const int elast = max(ne - 1, 0);

    // the rows of EU consecutive out-edges, all loads issued together; edge scalars are wave-uniform (SGPRs)
    auto load_group = [&](auto& q, const int j, const int je) {
        constexpr int EU = sizeof(q.p) / sizeof(int);
#pragma unroll
        for (int t = 0; t < EU; ++t) {
            const int idx = min(max(min(j + t, je - 1) - e0, 0), elast);
            int v;
            if constexpr (STAGED) { v = uni(s_v[idx]); q.p[t] = uni(s_p[idx]); q.cnv[t] = uni(s_cn[idx]); q.g1v[t] = uni(s_g1[idx]); q.g2v[t] = uni(s_g2[idx]); }
            else { v = a.col_dst[e0 + idx]; q.p[t] = a.pos_out[e0 + idx]; q.cnv[t] = a.cn[v]; q.g1v[t] = a.da1[v]; q.g2v[t] = a.da2[v]; }
            q.al[t] = a.alpha[(long long)q.p[t] * a.H + hw];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                vload<4>(a.X + (long long)v * Kp + off[i], q.xv[t][i]);
                q.mv[t][i] = fb_keep<MASK>(a.mask, a.mask_ld, v, off[i]);
            }
        }
    };
    // d alpha_e (raw) and the alpha'-weighted accumulation for the EU edges of a group
    auto use_group = [&](auto& q, const int j, const int je, const float (&ft)[NI][4], const float (&dz)[NI][4], float (&acc)[NI][4]) {
        constexpr int EU = sizeof(q.p) / sizeof(int);
        float fd[EU];
#pragma unroll
        for (int t = 0; t < EU; ++t) {
            fd[t] = 1.f;
            if (j + t < je) {                                      // wave-uniform (and behind every load of the group): the clamped
                                                                   // duplicates that pad a short group cost no arithmetic
                fd[t] = (a.drop_p > 0.f) ? drop_factor(a.seed, (unsigned long long)q.p[t] * a.H + hw, a.drop_p, a.drop_scale) : 1.f;
                const float coef = q.al[t] * fd[t];
                const float sc = q.cnv[t] * a.fscale, s1 = q.g1v[t] * a.fscale, s2 = q.g2v[t] * a.fscale;     // (uniform: scalar ALU)
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    float w1[4], w2[4];
                    vload<4>(s_wa + off[i], w1);
                    vload<4>(s_wa + Kp + off[i], w2);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float tv = sc * dz[i][k] + s1 * w1[k] + s2 * w2[k];
                        const float lk = (q.xv[t][i][k] > 0.f) ? live[i] : a.act_slope * live[i];
                        const float dp = ((q.mv[t][i] >> k) & 1u) ? tv * lk : 0.f;
                        part = fmaf(dp, ft[i][k], part);
                        acc[i][k] = fmaf(coef, dp, acc[i][k]);
                    }
                }
                part = wave_sum(part);
                if constexpr (NWH > 1) {                           // a head spans NWH waves: combine their partial dot products
                    if (l == 0) s_dot[t][w] = part;
                } else {
                    if (l == 0) a.dal[(long long)q.p[t] * a.H + hw] = part * fd[t];
                }
            }
        }
        if constexpr (NWH > 1) {
            __syncthreads();
            if (l == 0 && (w % NWH) == 0) {
#pragma unroll
                for (int t = 0; t < EU; ++t) {
                    float tot = 0.f;
#pragma unroll
                    for (int x = 0; x < NWH; ++x) tot += s_dot[t][w + x];
                    if (j + t < je) a.dal[(long long)q.p[t] * a.H + hw] = tot * fd[t];
                }
            }
            __syncthreads();
        }
    };

    for (int u = u0; u < u1; ++u) {
        const int un = u - u0;                                      // per-node scalars were staged with the edge scalars
        const int g = uni(s_ni[4 * un]), jb = uni(s_ni[4 * un + 1]), je = uni(s_ni[4 * un + 2]), pu = s_ni[4 * un + 3];
        const float g1u = uni(s_nf[4 * un]), g2u = uni(s_nf[4 * un + 1]), cnu = s_nf[4 * un + 2];
        float ft[NI][4], dz[NI][4], acc[NI][4];
        float xu[NI][4], xt[4], dzt[4];
        unsigned mu[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {                              // this node's own rows ...
            vload<4>(a.Y + (long long)u * a.ld_y + off[i], ft[i]);
            vload<4>(a.dZ + (long long)g * Kp + off[i], dz[i]);
            vload<4>(a.X + (long long)u * Kp + off[i], xu[i]);
            mu[i] = fb_keep<MASK>(a.mask, a.mask_ld, u, off[i]);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
        }
        vload<4>(a.X + (long long)u * Kp + tc, xt);
        vload<4>(a.dZ + (long long)g * Kp + tc, dzt);
        const unsigned mt = fb_keep<MASK>(a.mask, a.mask_ld, u, tc);
        FbGroup<NI, 2> q;
        load_group(q, jb, je);                                      // ... and its first two out-edges' rows: one round trip
        // own-row leftovers of cl_bwd_dx: d_wa partials (per workgroup, in LDS), position columns of d_X'
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float a1[4], a2[4];
            vload<4>(s_acc + off[i], a1);
            vload<4>(s_acc + Kp + off[i], a2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xd = ((mu[i] >> k) & 1u) ? xu[i][k] * a.fscale * live[i] : 0.f;
                a1[k] = fmaf(g1u, xd, a1[k]);
                a2[k] = fmaf(g2u, xd, a2[k]);
            }
            if (l + 64 * i < nvec) { vstore<4>(s_acc + off[i], a1); vstore<4>(s_acc + Kp + off[i], a2); }
        }
        if (tail) {
            float a1[4], a2[4], wt1[4], wt2[4];
            vload<4>(s_acc + tc, a1);
            vload<4>(s_acc + Kp + tc, a2);
            vload<4>(s_wa + tc, wt1);
            vload<4>(s_wa + Kp + tc, wt2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool keep = ((mt >> k) & 1u) != 0u;
                const float xd = keep ? xt[k] * a.fscale : 0.f;
                a1[k] = fmaf(g1u, xd, a1[k]);
                a2[k] = fmaf(g2u, xd, a2[k]);
                const int pc = tc + k - a.Kh;                      // position column (d_X' there has no activation factor)
                if (pc >= 0 && pc < a.Pd) s_dp[pu * a.Pd + pc] += keep ? a.fscale * (cnu * dzt[k] + g1u * wt1[k] + g2u * wt2[k]) : 0.f;
            }
            vstore<4>(s_acc + tc, a1);
            vstore<4>(s_acc + Kp + tc, a2);
        }
        if (jb < je) use_group(q, jb, je, ft, dz, acc);
        for (int j = jb + 2; j < je; j += FB_EU) {                  // a hub's further out-edges, FB_EU rows per round trip
            FbGroup<NI, FB_EU> q4;
            load_group(q4, j, je);
            use_group(q4, j, je, ft, dz, acc);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (l + 64 * i < nvec) vstore<4>(a.d_Y + (long long)u * a.ld_dy + off[i], acc[i]);
    }
}

template <bool MASK, int NI, int NWH /* waves per head = 4 / H */>
// (three workgroups per CU = 168 VGPRs hold the sweep up to NI = 2 -- rows of up to 2,048 columns, the MAG shape; wider rows (SemEval:
//  2,400) spilled 77 registers per lane there: two workgroups per CU, 256 VGPRs)
__global__ __launch_bounds__(256, (NI >= 3) ? 2 : TXE_FB_OCC) void gat_fused_bwd_kernel(const FusedBwdArgs a) {
    __shared__ int s_v[FB_MAXE], s_p[FB_MAXE], s_ni[4 * FB_NODES];
    __shared__ float s_cn[FB_MAXE], s_g1[FB_MAXE], s_g2[FB_MAXE], s_nf[4 * FB_NODES];
    __shared__ float s_dot[4][4];
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // [2][Kp] folded attention rows | [2][Kp] their gradient partials |
    const int b = xcd_remap(blockIdx.x, gridDim.x);                 // [vocab][Pd] position-embedding gradient partials
    const int u0 = b * a.npw, u1 = min(a.n_nodes, u0 + a.npw);
    const int Kp = a.Kp;
    float* s_wa = s_dyn;
    float* s_acc = s_dyn + 2 * Kp;
    float* s_dp = s_dyn + 4 * Kp;
    const int e0 = a.rowptr_out[u0], ne = a.rowptr_out[u1] - e0;
    if (threadIdx.x == 0) { s_v[0] = u0; s_p[0] = 0; s_cn[0] = 0.f; s_g1[0] = 0.f; s_g2[0] = 0.f; }   // (a workgroup without out-edges)
    __syncthreads();
    for (int i = threadIdx.x; i < min(ne, FB_MAXE); i += 256) {
        const int v = a.col_dst[e0 + i];
        s_v[i] = v; s_p[i] = a.pos_out[e0 + i];
        s_cn[i] = a.cn[v]; s_g1[i] = a.da1[v]; s_g2[i] = a.da2[v];
    }
    if (threadIdx.x < u1 - u0) {
        const int u = u0 + threadIdx.x;
        s_ni[4 * threadIdx.x] = a.gid[u]; s_ni[4 * threadIdx.x + 1] = a.rowptr_out[u]; s_ni[4 * threadIdx.x + 2] = a.rowptr_out[u + 1];
        s_ni[4 * threadIdx.x + 3] = a.pos[u];                       // (a readable dummy when there are no position columns)
        s_nf[4 * threadIdx.x] = a.da1[u]; s_nf[4 * threadIdx.x + 1] = a.da2[u]; s_nf[4 * threadIdx.x + 2] = a.cn[u];
    }
    for (int i = threadIdx.x; i < a.vocab * a.Pd; i += 256) s_dp[i] = 0.f;
    for (int i = threadIdx.x * 4; i < 2 * Kp; i += 1024) {
        *reinterpret_cast<float4*>(s_wa + i) = *reinterpret_cast<const float4*>(a.wa + i);
        *reinterpret_cast<float4*>(s_acc + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (ne <= FB_MAXE) fb_body<MASK, NI, NWH, true>(a, b, u0, u1, e0, ne, s_v, s_p, s_cn, s_g1, s_g2, s_ni, s_nf, s_dot, s_dp, s_wa, s_acc);
    else fb_body<MASK, NI, NWH, false>(a, b, u0, u1, e0, ne, s_v, s_p, s_cn, s_g1, s_g2, s_ni, s_nf, s_dot, s_dp, s_wa, s_acc);
    // per-workgroup partials: folded attention rows' gradient [2][Kp], position-embedding gradient [vocab][Pd]
    __syncthreads();
    float* dw = a.dwa_part + (long long)b * 2 * Kp;
    for (int i = threadIdx.x * 4; i < 2 * Kp; i += 1024) *reinterpret_cast<float4*>(dw + i) = *reinterpret_cast<const float4*>(s_acc + i);
    for (int i = threadIdx.x; i < a.vocab * a.Pd; i += 256) a.ppart[(long long)b * a.vocab * a.Pd + i] = s_dp[i];
}

// ---- the same sweep, WALKING EGONETS (dataset.py:404-437: parents -> anchor, anchor -> siblings, self loops) -----------------------
// The sweep above fetches X'[v] once per out-edge (u -> v): an anchor's row once per parent, a sibling's row once for the anchor and
// once as its own -- 64 MB of re-fetched rows on the training batch (FETCH_SIZE 374 MB against 326 MB algorithmic).  In an egonet every
// edge that is not a self loop touches ONE node, the hub h (the anchor): parents u have out-edges {u, h}, siblings s have {s} and the
// in-edge h -> s.  With the hub's three row slices in registers -- ft[h], d_pre[h] and the accumulating d_ft[h] -- every other node's rows
// are read exactly once:
//     hub h        : d_pre[h], self edge
//     parent u     : d_pre[u], self edge;  edge u -> h: d alpha = <d_pre[h], ft[u]>, d_ft[u] += alpha' d_pre[h]
//     sibling s    : d_pre[s], self edge;  edge h -> s: d alpha = <d_pre[s], ft[h]>, d_ft[h] += alpha' d_pre[s]
// X', Y are streamed once, d_Y written once: the algorithmic bytes.
// Work list: the nodes of a hub-shaped graph in the order hub, parents, siblings (list position = node index except inside a graph);
// a workgroup walks npw consecutive LIST POSITIONS, two per round trip -- every workgroup the same amount of work, whatever the graph
// sizes (a 54-node egonet beside 2-node ones).  A graph cut by a workgroup boundary: the later workgroup first loads the hub's rows
// again (d_pre[h], ft[h]; nothing written), and leaves ITS share of d_ft[h] in hpart[workgroup]; gat_attn_bwd_reduce_a_kernel -- the next
// launch -- adds those rows to d_Y[h] in workgroup order (fused_hub_fixup_job: deterministic, no atomics).
// The shape is CHECKED per graph from the CSR arrays (out-degrees, out-lists of the small nodes, in-lists of the siblings -- never the
// position labels), by every workgroup that touches the graph: a graph that is not hub-shaped -- or has more than EGO_MAXN nodes -- is
// walked by the generic body above (fb_body, edge scalars from global memory) for the source nodes in the workgroup's window.
// One head per wave (H = 4: the per-head dot products are wave-local); other head counts keep the kernel above.
constexpr int EGO_MAXN = 64;                             // largest hub-shaped graph walked from registers
constexpr int EGO_TAB = FB_NODES + 2 * EGO_MAXN;         // nodes of the graphs that intersect a window of <= FB_NODES positions
enum { EGO_SKIP = 0, EGO_HUB = 1, EGO_PRE = 2, EGO_POST = 3, EGO_FOREIGN = 4 };

// list position (local index t inside a hub-shaped graph with hub h) -> local node index
__device__ __forceinline__ int ego_node_of(int t, int h) { return t == 0 ? h : (t <= h ? t - 1 : t); }

// The walk plan of a batch: what the staging phases (1)-(3) of the kernel below work out per workgroup and step -- hub, roles, CSR
// positions, the list order -- depends on the graphs alone, so it can be done ONCE per batch (it is a view of the graph like the two CSR
// orders).  8 ints per LIST POSITION p: the node walked there, flags (role | walkable << 4 | at most EGO_MAXN nodes << 5), the destination
// CSR positions of its self loop and of its edge with the hub, the graph's hub (node id), the graph's first position.  With a plan the
// sweep's staging is two trips (plan; then the per-node scalars and the edge coefficients) instead of eight.  One wave per graph.
constexpr int EGO_PLAN_W = 8;
__global__ __launch_bounds__(256) void egonet_walk_plan_kernel(const int* __restrict__ rowptr_in, const int* __restrict__ col_src,
                                                               const int* __restrict__ rowptr_out, const int* __restrict__ col_dst,
                                                               const int* __restrict__ pos_out, const int* __restrict__ goff, const int G,
                                                               int* __restrict__ plan) {
    const int g = (int)(((long long)blockIdx.x * 256 + threadIdx.x) >> 6), l = threadIdx.x & 63;
    if (g >= G) return;
    const int o = goff[g], n = goff[g + 1] - o;
    auto put = [&](int p, int node, int flags, int ps, int ph, int hub) {
        int4* q = reinterpret_cast<int4*>(plan + (long long)p * EGO_PLAN_W);
        q[0] = make_int4(node, flags, ps, ph);
        q[1] = make_int4(hub, o, 0, 0);
    };
    if (n > EGO_MAXN) {                                             // never walked from registers: list position = node
        for (int i = l; i < n; i += 64) put(o + i, o + i, EGO_SKIP, 0, 0, -1);
        return;
    }
    if (n == 0) return;
    const bool act = l < n;
    const int v = o + (act ? l : 0);
    const int e0 = rowptr_out[v], d = act ? rowptr_out[v + 1] - e0 : 0;
    int tgt = -1;
    if (d == 2) { const int d0 = col_dst[e0], d1 = col_dst[e0 + 1]; tgt = ((d0 == v) ? d1 : d0) - o; }
    // the hub: THE node of out-degree >= 3, else the target of the first node of out-degree 2, else node 0 of a single-node graph
    const unsigned long long mbig = __ballot(d >= 3), m2 = __ballot(d == 2);
    int h = -1;
    if (mbig != 0ull) h = __ffsll((long long)mbig) - 1;
    else if (m2 != 0ull) h = __shfl(tgt, __ffsll((long long)m2) - 1, 64);
    else if (n == 1) h = 0;
    bool gok = __popcll(mbig) <= 1 && h >= 0 && h < n;
    int role = EGO_SKIP, pself = -1, phub = -1;
    if (gok) {
        const int vh = o + h;
        const int n_post = __popcll(__ballot(act && l != h && d == 1));
        bool ok = true;
        if (act) {
            const int pi0 = rowptr_in[v], din = rowptr_in[v + 1] - pi0;
            if (l == h) {                          // hub: itself in its in-list; out-degree = 1 + #siblings (the siblings check their side)
                role = EGO_HUB;
                for (int q = 0; q < din; ++q) if (col_src[pi0 + q] == v) pself = pi0 + q;
                ok = pself >= 0 && d == 1 + n_post;
                phub = pself;
            } else if (d == 2) {                   // parent: out-list {self, hub}
                role = EGO_PRE;
                const int d0 = col_dst[e0], d1 = col_dst[e0 + 1];
                if (d0 == v && d1 == vh) { pself = pos_out[e0]; phub = pos_out[e0 + 1]; }
                else if (d1 == v && d0 == vh) { pself = pos_out[e0 + 1]; phub = pos_out[e0]; }
                else ok = false;
            } else if (d == 1) {                   // sibling: in-list {hub, self}; its one out-edge is then the self loop
                role = EGO_POST;
                if (din == 2) {
                    const int s0 = col_src[pi0], s1 = col_src[pi0 + 1];
                    if (s0 == v && s1 == vh) { pself = pi0; phub = pi0 + 1; }
                    else if (s1 == v && s0 == vh) { pself = pi0 + 1; phub = pi0; }
                    else ok = false;
                } else ok = false;
            } else ok = false;
        }
        gok = __ballot(act && !ok) == 0ull;
    }
    if (!act) return;
    if (gok) put(o + ((l == h) ? 0 : (l < h ? l + 1 : l)), v, role | 16 | 32, max(pself, 0), max(phub, 0), o + h);
    else put(o + l, v, EGO_SKIP | 32, 0, 0, -1);
}

#ifndef TXE_EGO_OCC
#define TXE_EGO_OCC 3
#endif
#ifndef TXE_EGO_SLOTS
#define TXE_EGO_SLOTS 1
#endif
template <bool MASK, int NI>
__global__ __launch_bounds__(256, (NI >= 3) ? 2 : TXE_EGO_OCC) void gat_fused_bwd_ego_kernel(const FusedBwdArgs a) {
    __shared__ int s_v[4], s_p[4], s_ni[4 * FB_NODES];                         // (the generic body's per-node table; its edge tables are not used)
    __shared__ float s_cn[4], s_g1[4], s_g2[4], s_nf[4 * FB_NODES];
    __shared__ float s_dot[4][4];
    // per list position of the window (+ one entry for a foreign hub, + one skip entry that pads an odd count)
    __shared__ int t_node[FB_NODES + 2], t_role[FB_NODES + 2], t_self[FB_NODES + 2], t_hub[FB_NODES + 2], t_dz[FB_NODES + 2], t_pos[FB_NODES + 2];
    __shared__ float t_cn[FB_NODES + 2], t_g1[FB_NODES + 2], t_g2[FB_NODES + 2];
    // per position and head: alpha' = alpha * dropout factor and the factor itself, of the self loop [0..3] and of the edge with the hub [4..7]
    __shared__ float t_coef[FB_NODES + 2][8], t_fd[FB_NODES + 2][8];
    // per node of the intersecting graphs (staging)
    __shared__ int n_deg[EGO_TAB], n_tgt[EGO_TAB], n_role[EGO_TAB], n_self[EGO_TAB], n_hubp[EGO_TAB];
    __shared__ int g_hub[FB_NODES], g_ok[FB_NODES];
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int Kp = a.Kp;
    float* s_wa = s_dyn;
    float* s_acc = s_dyn + 2 * Kp;
    float* s_dp = s_dyn + 4 * Kp;
    const int tid = threadIdx.x;
    // ---- the window of list positions and the graphs that intersect it ----
    const int u0 = b * a.npw, u1 = min(a.n_nodes, u0 + a.npw), nw = u1 - u0;   // (nw >= 1: the grid has ceil(n / npw) workgroups)
    __shared__ int t_ok[FB_NODES + 2], t_gs[FB_NODES + 2];                     // (with a plan) the position's graph is walked; its first position
    const bool planned = a.plan != nullptr;
    int gF = 0, gL = 0, ng = 0, offF = 0, endL = 0, tb = 0, te = 0;
    if (!planned) {
        gF = a.ggid[u0]; gL = a.ggid[u1 - 1]; ng = gL - gF + 1;                // <= npw <= FB_NODES graphs
        offF = a.goff[gF]; endL = a.goff[gL + 1];
        tb = (a.goff[gF + 1] - offF <= EGO_MAXN) ? offF : u0;                 // first / one-past-last node with a staging entry
        te = (endL - a.goff[gL] <= EGO_MAXN) ? endL : u1;                      // (only the first and the last graph reach outside the window)
    }
    // (with a plan the entries 0..nw are written whole by the staging loop below: no barrier between defaults and values)
    for (int i = tid; i < FB_NODES + 2; i += 256)
        if (!planned || i > nw) { t_node[i] = u0; t_role[i] = EGO_SKIP; t_self[i] = 0; t_hub[i] = 0; t_dz[i] = 0; t_pos[i] = 0; t_cn[i] = 0.f; t_g1[i] = 0.f; t_g2[i] = 0.f; }
    for (int i = tid; i < (FB_NODES + 2) * 8; i += 256)
        if (!planned || (i >> 3) > nw) { t_coef[i >> 3][i & 7] = 0.f; t_fd[i >> 3][i & 7] = 0.f; }
    for (int i = tid; i < a.vocab * a.Pd; i += 256) s_dp[i] = 0.f;
    for (int i = tid * 4; i < 2 * Kp; i += 1024) {
        *reinterpret_cast<float4*>(s_wa + i) = *reinterpret_cast<const float4*>(a.wa + i);
        *reinterpret_cast<float4*>(s_acc + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bool owes_hpart = false;
    if (planned) {
        // ---- staging from the batch's walk plan: trip 1 = the plan entries (the window's first one with them: does the window start
        //      inside a graph?), trip 2 = the nodes' scalars and the edge coefficients; one barrier ----
        const int4 p0 = *reinterpret_cast<const int4*>(a.plan + (long long)u0 * EGO_PLAN_W);
        const int4 p0b = *reinterpret_cast<const int4*>(a.plan + (long long)u0 * EGO_PLAN_W + 4);
        for (int i = tid; i < (nw + 1) * 8; i += 256) {
            const int t = i >> 3, e = (i >> 2) & 1, hd = i & 3;
            const long long pp = (long long)(u0 + (t < nw ? t : 0)) * EGO_PLAN_W;
            const int4 q = *reinterpret_cast<const int4*>(a.plan + pp);
            const int4 qb = *reinterpret_cast<const int4*>(a.plan + pp + 4);
            const bool foreign = p0b.y < u0 && (p0.y & 16) != 0;               // the window starts inside a graph that is walked
            const int role = (t < nw) ? (q.y & 15) : (foreign ? EGO_FOREIGN : EGO_SKIP);
            const int v = (role == EGO_SKIP) ? u0 : ((t < nw) ? q.x : qb.x);
            float fd = 0.f, cf = 0.f;
            if (role != EGO_SKIP) {
                const long long idx = (long long)(e ? q.w : q.z) * a.H + hd;   // (a foreign hub's coefficients are never used)
                fd = (a.drop_p > 0.f) ? drop_factor(a.seed, (unsigned long long)idx, a.drop_p, a.drop_scale) : 1.f;
                cf = (t < nw) ? a.alpha[idx] * fd : 0.f;
            }
            t_fd[t][i & 7] = fd;
            t_coef[t][i & 7] = cf;
            if ((i & 7) == 0) {
                const bool live = role != EGO_SKIP;
                t_ok[t] = (q.y >> 4) & 1; t_gs[t] = qb.y;
                t_node[t] = v; t_role[t] = role; t_self[t] = live ? q.z : 0; t_hub[t] = live ? q.w : 0;
                t_dz[t] = live ? a.gid[v] : 0; t_pos[t] = live ? a.pos[v] : 0;
                t_cn[t] = live ? a.cn[v] : 0.f; t_g1[t] = live ? a.da1[v] : 0.f; t_g2[t] = live ? a.da2[v] : 0.f;
            }
        }
        owes_hpart = p0b.y < u0 && (p0.y & 32) != 0;                           // (a graph of at most EGO_MAXN nodes, walked or not)
        __syncthreads();
    } else {
    if (ng > FB_NODES) {
        // more graphs than positions in the window: it holds EMPTY graphs (an egonet has at least its anchor) -- not a batch of egonets;
        // every source node of the window through the generic body, and the row a fix-up pass may read cleared
        __syncthreads();
        if (tid < nw) {
            const int u = u0 + tid;
            s_ni[4 * tid] = a.gid[u]; s_ni[4 * tid + 1] = a.rowptr_out[u]; s_ni[4 * tid + 2] = a.rowptr_out[u + 1];
            s_ni[4 * tid + 3] = a.pos[u];
            s_nf[4 * tid] = a.da1[u]; s_nf[4 * tid + 1] = a.da2[u]; s_nf[4 * tid + 2] = a.cn[u];
        }
        __syncthreads();
        const int e0 = a.rowptr_out[u0], ne = a.rowptr_out[u1] - e0;
        fb_body<MASK, NI, 1, false>(a, b, u0, u1, e0, ne, s_v, s_p, s_cn, s_g1, s_g2, s_ni, s_nf, s_dot, s_dp, s_wa, s_acc);
        if (u0 > offF && a.goff[gF + 1] - offF <= EGO_MAXN)
            for (int c = tid; c < a.H * a.D; c += 256) a.hpart[(long long)b * a.H * a.D + c] = 0.f;
        __syncthreads();
        float* dwg = a.dwa_part + (long long)b * 2 * Kp;
        for (int i = tid * 4; i < 2 * Kp; i += 1024) *reinterpret_cast<float4*>(dwg + i) = *reinterpret_cast<const float4*>(s_acc + i);
        for (int i = tid; i < a.vocab * a.Pd; i += 256) a.ppart[(long long)b * a.vocab * a.Pd + i] = s_dp[i];
        return;
    }
    if (tid < ng) { g_ok[tid] = (a.goff[gF + tid + 1] - a.goff[gF + tid] <= EGO_MAXN) ? 1 : 0; g_hub[tid] = 0; }
    __syncthreads();
    // (1) out-degree, and the non-self target of a node of out-degree 2 -- one thread per node of the graphs that fit
    int my_g = -1, my_i = 0, my_n = 0, my_v = 0, my_base = 0;
    if (tid < te - tb) {
        const int v = tb + tid, g = a.ggid[v];
        if (g_ok[g - gF]) { my_g = g - gF; my_base = a.goff[g] - tb; my_i = v - a.goff[g]; my_n = a.goff[g + 1] - a.goff[g]; my_v = v; }
    }
    if (my_g >= 0) {
        const int e0 = a.rowptr_out[my_v], d = a.rowptr_out[my_v + 1] - e0;
        n_deg[tid] = d;
        int tgt = -1;
        if (d == 2) { const int d0 = a.col_dst[e0], d1 = a.col_dst[e0 + 1]; tgt = (d0 == my_v) ? d1 : d0; }
        n_tgt[tid] = tgt - (tb + my_base);                                 // local index inside the graph (or out of range)
    }
    __syncthreads();
    // (2) the hub of every graph: THE node of out-degree >= 3, else the target of the first node of out-degree 2, else node 0 of a
    //     single-node graph
    if (tid < ng && g_ok[tid]) {
        const int base = a.goff[gF + tid] - tb, n = a.goff[gF + tid + 1] - a.goff[gF + tid];
        int h = -1, big = 0;
        for (int i = 0; i < n; ++i) if (n_deg[base + i] >= 3) { h = i; ++big; }
        if (big == 0) {
            for (int i = 0; i < n && h < 0; ++i) if (n_deg[base + i] == 2) h = n_tgt[base + i];
            if (h < 0) h = (n == 1) ? 0 : -1;
        }
        if (big > 1 || h < 0 || h >= n) g_ok[tid] = 0; else g_hub[tid] = h;
    }
    __syncthreads();
    // (3) every node against the hub shape; its role and the destination-CSR positions of its self loop and of its edge with the hub
    if (my_g >= 0 && g_ok[my_g]) {
        const int h = g_hub[my_g], vh = tb + my_base + h, d = n_deg[tid];
        const int pi0 = a.rowptr_in[my_v], din = a.rowptr_in[my_v + 1] - pi0;
        int role = EGO_SKIP, pself = -1, phub = -1;
        bool ok = true;
        if (my_i == h) {                       // hub: itself in its in-list; out-degree = 1 + #siblings (the siblings check their side)
            role = EGO_HUB;
            for (int q = 0; q < din; ++q) if (a.col_src[pi0 + q] == my_v) pself = pi0 + q;
            int n_post = 0;
            for (int i = 0; i < my_n; ++i) n_post += (i != h && n_deg[my_base + i] == 1) ? 1 : 0;
            ok = pself >= 0 && d == 1 + n_post;
            phub = pself;
        } else if (d == 2) {                   // parent: out-list {self, hub}
            role = EGO_PRE;
            const int e0 = a.rowptr_out[my_v];
            const int d0 = a.col_dst[e0], d1 = a.col_dst[e0 + 1];
            if (d0 == my_v && d1 == vh) { pself = a.pos_out[e0]; phub = a.pos_out[e0 + 1]; }
            else if (d1 == my_v && d0 == vh) { pself = a.pos_out[e0 + 1]; phub = a.pos_out[e0]; }
            else ok = false;
        } else if (d == 1) {                   // sibling: in-list {hub, self}; its one out-edge is then the self loop
            role = EGO_POST;
            if (din == 2) {
                const int s0 = a.col_src[pi0], s1 = a.col_src[pi0 + 1];
                if (s0 == my_v && s1 == vh) { pself = pi0; phub = pi0 + 1; }
                else if (s1 == my_v && s0 == vh) { pself = pi0 + 1; phub = pi0; }
                else ok = false;
            } else ok = false;
        } else ok = false;
        if (!ok) g_ok[my_g] = 0;                 // (benign race: every writer stores 0)
        n_role[tid] = role; n_self[tid] = max(pself, 0); n_hubp[tid] = max(phub, 0);
    }
    __syncthreads();
    // (4) the window's list positions -> table entries; entry nw: the hub of a graph whose list the window enters in the middle
    if (tid <= nw) {
        int v = -1, role = EGO_SKIP, idx = 0;
        if (tid < nw) {
            const int p = u0 + tid, g = a.ggid[p];
            if (g_ok[g - gF]) { v = a.goff[g] + ego_node_of(p - a.goff[g], g_hub[g - gF]); idx = v - tb; role = n_role[idx]; }
        } else if (g_ok[0] && u0 > offF) { v = offF + g_hub[0]; idx = v - tb; role = EGO_FOREIGN; }
        if (v >= 0) {
            t_node[tid] = v; t_role[tid] = role; t_self[tid] = n_self[idx]; t_hub[tid] = n_hubp[idx];
            t_dz[tid] = a.gid[v]; t_pos[tid] = a.pos[v];
            t_cn[tid] = a.cn[v]; t_g1[tid] = a.da1[v]; t_g2[tid] = a.da2[v];
        }
    }
    __syncthreads();
    // (5) the edge scalars of every (position, edge, head): one thread each -- the walk reads them from LDS
    for (int i = tid; i < (nw + 1) * 8; i += 256) {
        const int t = i >> 3, e = (i >> 2) & 1, hd = i & 3;
        if (t_role[t] != EGO_SKIP) {
            const long long idx = (long long)(e ? t_hub[t] : t_self[t]) * a.H + hd;
            const float fd = (a.drop_p > 0.f) ? drop_factor(a.seed, (unsigned long long)idx, a.drop_p, a.drop_scale) : 1.f;
            t_fd[t][i & 7] = fd;
            t_coef[t][i & 7] = a.alpha[idx] * fd;
        }
    }
    __syncthreads();

    }   // (!planned)

    const int w = uni((int)(tid >> 6)), l = tid & 63;
    const int F = a.H * a.D, SL = F >> 2, nvec = SL >> 2;
    const int c0 = w * SL, hw = c0 / a.D;
    int off[NI];
    float live[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = l + 64 * i;
        off[i] = c0 + 4 * ((j < nvec) ? j : 0);
        live[i] = (j < nvec) ? 1.f : 0.f;
    }
    const int tvec = (Kp - F) >> 2;
    const bool tail = (w == 0) && (l < tvec);
    const int tc = min(F + 4 * (tail ? l : 0), Kp - 4);
    float fth[NI][4], pph[NI][4], acch[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) { fth[i][k] = 0.f; pph[i][k] = 0.f; acch[i][k] = 0.f; }
    int hub_node = -1;                           // the hub whose slices are in registers; hub_home: its d_ft goes to d_Y (else to hpart[b])
    bool hub_home = true;
    // a window that starts inside a graph of <= EGO_MAXN nodes owes the fix-up pass a row hpart[b]: its share of the hub's d_ft, or
    // zeros if the graph turned out not to be hub-shaped (fused_hub_fixup_job repeats only the cheap half of the shape check)
    if (!planned) owes_hpart = u0 > offF && (a.goff[gF + 1] - offF <= EGO_MAXN);
    bool paid_hpart = false;
    auto flush_hub = [&]() {
        if (hub_node >= 0) {
            if (!hub_home) paid_hpart = true;
            float* dst = hub_home ? a.d_Y + (long long)hub_node * a.ld_dy : a.hpart + (long long)b * F;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (l + 64 * i < nvec) vstore<4>(dst + off[i], acch[i]);
        }
    };

    // the walk, NS entries per round trip: entry nw first if it is a foreign hub, then the positions
    constexpr int NS = TXE_EGO_SLOTS;
    for (int t0 = (t_role[nw] == EGO_FOREIGN) ? -1 : 0; t0 < nw; t0 += NS) {
        // ---- every load of the two entries first (rows, masks, edge scalars), nothing in between ----
        int vv[NS], role[NS], ps[NS], ph[NS], pv[NS];
        float cnv[NS], g1v[NS], g2v[NS], cfs[NS], fds[NS], cfh[NS], fdh2[NS];
        float ft[NS][NI][4], xv[NS][NI][4], dz[NS][NI][4], xt[NS][4], dzt[NS][4];
        unsigned mv[NS][NI], mt[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int t = (t0 + q < 0) ? nw : ((t0 + q < nw) ? t0 + q : FB_NODES + 1);     // (FB_NODES + 1: an entry that stays EGO_SKIP)
            vv[q] = uni(t_node[t]); role[q] = uni(t_role[t]); ps[q] = uni(t_self[t]); ph[q] = uni(t_hub[t]); pv[q] = t_pos[t];
            cnv[q] = uni(t_cn[t]); g1v[q] = uni(t_g1[t]); g2v[q] = uni(t_g2[t]);
            const int dzr = uni(t_dz[t]);
            cfs[q] = uni(t_coef[t][hw]); fds[q] = uni(t_fd[t][hw]); cfh[q] = uni(t_coef[t][4 + hw]); fdh2[q] = uni(t_fd[t][4 + hw]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                vload<4>(a.Y + (long long)vv[q] * a.ld_y + off[i], ft[q][i]);
                vload<4>(a.X + (long long)vv[q] * Kp + off[i], xv[q][i]);
                vload<4>(a.dZ + (long long)dzr * Kp + off[i], dz[q][i]);
                mv[q][i] = fb_keep<MASK>(a.mask, a.mask_ld, vv[q], off[i]);
            }
            vload<4>(a.X + (long long)vv[q] * Kp + tc, xt[q]);
            vload<4>(a.dZ + (long long)dzr * Kp + tc, dzt[q]);
            mt[q] = fb_keep<MASK>(a.mask, a.mask_ld, vv[q], tc);
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (role[q] == EGO_SKIP) continue;                        // (wave-uniform)
            const int v = vv[q];
            const bool own = role[q] != EGO_FOREIGN;                  // a foreign hub: d_pre and ft only -- its own-row work belongs to its home
            const float sc = cnv[q] * a.fscale, s1 = g1v[q] * a.fscale, s2 = g2v[q] * a.fscale;
            float dp[NI][4], acc[NI][4];
            const float fd = fds[q], coef = cfs[q];
            const float go1 = own ? g1v[q] : 0.f, go2 = own ? g2v[q] : 0.f;
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float w1[4], w2[4], a1[4], a2[4];
                vload<4>(s_wa + off[i], w1);
                vload<4>(s_wa + Kp + off[i], w2);
                vload<4>(s_acc + off[i], a1);
                vload<4>(s_acc + Kp + off[i], a2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool keep = ((mv[q][i] >> k) & 1u) != 0u;
                    const float tv = sc * dz[q][i][k] + s1 * w1[k] + s2 * w2[k];
                    const float lk = (xv[q][i][k] > 0.f) ? live[i] : a.act_slope * live[i];
                    dp[i][k] = keep ? tv * lk : 0.f;
                    part = fmaf(dp[i][k], ft[q][i][k], part);
                    acc[i][k] = coef * dp[i][k];
                    const float xd = keep ? xv[q][i][k] * a.fscale * live[i] : 0.f;       // own-row leftovers of cl_bwd_dx: d_wa partials
                    a1[k] = fmaf(go1, xd, a1[k]);
                    a2[k] = fmaf(go2, xd, a2[k]);
                }
                if (l + 64 * i < nvec) { vstore<4>(s_acc + off[i], a1); vstore<4>(s_acc + Kp + off[i], a2); }
            }
            if (tail && own) {
                float a1[4], a2[4], wt1[4], wt2[4];
                vload<4>(s_acc + tc, a1);
                vload<4>(s_acc + Kp + tc, a2);
                vload<4>(s_wa + tc, wt1);
                vload<4>(s_wa + Kp + tc, wt2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool keep = ((mt[q] >> k) & 1u) != 0u;
                    const float xd = keep ? xt[q][k] * a.fscale : 0.f;
                    a1[k] = fmaf(g1v[q], xd, a1[k]);
                    a2[k] = fmaf(g2v[q], xd, a2[k]);
                    const int pc = tc + k - a.Kh;                      // position column (d_X' there has no activation factor)
                    if (pc >= 0 && pc < a.Pd) s_dp[pv[q] * a.Pd + pc] += keep ? a.fscale * (cnv[q] * dzt[q][k] + g1v[q] * wt1[k] + g2v[q] * wt2[k]) : 0.f;
                }
                vstore<4>(s_acc + tc, a1);
                vstore<4>(s_acc + Kp + tc, a2);
            }
            if (own) {
                part = wave_sum(part);
                if (l == 0) a.dal[(long long)ps[q] * a.H + hw] = part * fd;           // the self loop's raw d alpha
            }
            if (role[q] == EGO_HUB || role[q] == EGO_FOREIGN) {
                flush_hub();
                hub_node = v; hub_home = own;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { fth[i][k] = ft[q][i][k]; pph[i][k] = dp[i][k]; acch[i][k] = own ? acc[i][k] : 0.f; }
            } else {
                const float fdh = fdh2[q], coefh = cfh[q];
                float part2 = 0.f;
                if (role[q] == EGO_PRE) {          // edge v -> hub: d_pre[hub] against this node's ft, accumulated into this node's d_ft
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) { part2 = fmaf(pph[i][k], ft[q][i][k], part2); acc[i][k] = fmaf(coefh, pph[i][k], acc[i][k]); }
                } else {                           // edge hub -> v: this node's d_pre against the hub's ft, accumulated into the hub's d_ft
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) { part2 = fmaf(dp[i][k], fth[i][k], part2); acch[i][k] = fmaf(coefh, dp[i][k], acch[i][k]); }
                }
                part2 = wave_sum(part2);
                if (l == 0) a.dal[(long long)ph[q] * a.H + hw] = part2 * fdh;
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (l + 64 * i < nvec) vstore<4>(a.d_Y + (long long)v * a.ld_dy + off[i], acc[i]);
            }
        }
    }
    flush_hub();
    if (owes_hpart && !paid_hpart) {
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (l + 64 * i < nvec) vstore<4>(a.hpart + (long long)b * F + off[i], z);
    }
    // ---- graphs that are not hub-shaped (or too large): the generic body over their source nodes inside the window ----
    for (int gi = 0, tp = 0; planned ? tp < nw : gi < ng; ++gi) {
        int c, cu1;
        if (planned) {                                               // the next stretch of positions of ONE graph that is not walked
            if (t_ok[tp]) { ++tp; continue; }                        // (LDS values: the same for every thread)
            const int gs = t_gs[tp];
            c = u0 + tp;
            while (tp < nw && !t_ok[tp] && t_gs[tp] == gs) ++tp;
            cu1 = u0 + tp;
        } else {
        if (g_ok[gi]) continue;                                      // (LDS value: the same for every thread)
        c = max(u0, a.goff[gF + gi]); cu1 = min(u1, a.goff[gF + gi + 1]);
        }
        __syncthreads();
        if (tid < cu1 - c) {
            const int u = c + tid;
            s_ni[4 * tid] = a.gid[u]; s_ni[4 * tid + 1] = a.rowptr_out[u]; s_ni[4 * tid + 2] = a.rowptr_out[u + 1];
            s_ni[4 * tid + 3] = a.pos[u];
            s_nf[4 * tid] = a.da1[u]; s_nf[4 * tid + 1] = a.da2[u]; s_nf[4 * tid + 2] = a.cn[u];
        }
        __syncthreads();
        const int e0 = a.rowptr_out[c], ne = a.rowptr_out[cu1] - e0;
        fb_body<MASK, NI, 1, false>(a, b, c, cu1, e0, ne, s_v, s_p, s_cn, s_g1, s_g2, s_ni, s_nf, s_dot, s_dp, s_wa, s_acc);
    }
    __syncthreads();
    float* dw = a.dwa_part + (long long)b * 2 * Kp;
    for (int i = tid * 4; i < 2 * Kp; i += 1024) *reinterpret_cast<float4*>(dw + i) = *reinterpret_cast<const float4*>(s_acc + i);
    for (int i = tid; i < a.vocab * a.Pd; i += 256) a.ppart[(long long)b * a.vocab * a.Pd + i] = s_dp[i];
}

// A hub-shaped graph cut by workgroup boundaries of the walk above: d_Y[hub] (written by the hub's home workgroup) += the later
// workgroups' shares, in workgroup order.  Block j stands for the boundary in front of window j; it acts only if that boundary cuts a
// graph whose hub lives in window j - 1... or earlier but this is the FIRST boundary inside the graph -- every cut graph is fixed once.
struct HubFixArgs { const int *goff, *ggid, *rowptr_out, *col_dst; int n_nodes, npw, nblocks, F; const float* hpart; float* d_Y; long long ld_dy; };
__device__ __forceinline__ void fused_hub_fixup_job(const int j, const HubFixArgs& a) {
    const int p = j * a.npw;                                         // first list position of window j (1 <= j < nblocks)
    const int g = a.ggid[p], o = a.goff[g], n = a.goff[g + 1] - o;
    if (o == p || n > EGO_MAXN) return;                              // no graph is cut here / never hub-walked
    const int bh = o / a.npw;                                        // home window of the hub (list position o)
    if (j != bh + 1) return;                                         // (the first boundary inside the graph does the whole job)
    // the graph's hub and whether it was hub-walked at all: the same rule as the sweep (out-degrees; the full shape check is repeated
    // cheaply: a graph that failed there wrote no hpart rows and must not be touched -- recompute the verdict)
    __shared__ int s_h, s_ok;
    if (threadIdx.x == 0) {
        int h = -1, big = 0;
        for (int i = 0; i < n; ++i) if (a.rowptr_out[o + i + 1] - a.rowptr_out[o + i] >= 3) { h = i; ++big; }
        if (big == 0) {
            for (int i = 0; i < n && h < 0; ++i) {
                const int e0 = a.rowptr_out[o + i];
                if (a.rowptr_out[o + i + 1] - e0 == 2) { const int d0 = a.col_dst[e0], d1 = a.col_dst[e0 + 1]; h = ((d0 == o + i) ? d1 : d0) - o; }
            }
            if (h < 0) h = (n == 1) ? 0 : -1;
        }
        s_h = h; s_ok = (big <= 1 && h >= 0 && h < n) ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok) return;
    const int bl = (o + n - 1) / a.npw;
    float* dst = a.d_Y + (long long)(o + s_h) * a.ld_dy;
    for (int c = threadIdx.x; c < a.F; c += 256) {
        float v = dst[c];
        for (int bb = bh + 1; bb <= bl; ++bb) v += a.hpart[(long long)bb * a.F + c];
        dst[c] = v;
    }
}

// Softmax + leaky-relu backward of a GATLayer's attention from the raw d alpha of the fused sweep, edge level:
//   dz_p = alpha_p (dal_p - sum_q alpha_q dal_q) leaky'(a_src[u_p] + a_dst[v]);  d a_dst[v] = sum_in dz;  d a_src[u] = sum_out dz
// written into the a1 / a2 columns of d_Y (and zeros into its padding columns).  A workgroup owns FA_GRAPHS consecutive graphs:
// the edges of a batched graph stay inside it, so the destination-side and source-side halves only need a workgroup barrier.
constexpr int FA_GRAPHS = 8;
constexpr int FA_LIGHT = 8;         // degrees up to this are walked by one thread per (node, head); heavier nodes by a whole wave
struct AttnBwdArgs {
    const int *rowptr_in, *col_src, *rowptr_out, *pos_out, *graph_off;
    int G;
    const float* Y; long long ld_y; int H, F; float slope;
    const float *alpha, *dal;
    float *dz, *d_Y; long long ld_dy; int n_pad;
};
__device__ __forceinline__ void gat_attn_bwd_job(const int bid, const AttnBwdArgs& a) {
    const int* __restrict__ rowptr_in = a.rowptr_in; const int* __restrict__ col_src = a.col_src;
    const int* __restrict__ rowptr_out = a.rowptr_out; const int* __restrict__ pos_out = a.pos_out;
    const int* __restrict__ graph_off = a.graph_off; const int G = a.G;
    const float* __restrict__ Y = a.Y; const long long ld_y = a.ld_y; const int H = a.H, F = a.F; const float slope = a.slope;
    const float* __restrict__ alpha = a.alpha; const float* __restrict__ dal = a.dal;
    float* __restrict__ dz = a.dz; float* __restrict__ d_Y = a.d_Y; const long long ld_dy = a.ld_dy; const int n_pad = a.n_pad;
    __shared__ int s_heavy[2][256], s_nh[2];                        // heavy destinations / sources found by the light passes
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int g0 = bid * FA_GRAPHS, g1 = min(G, g0 + FA_GRAPHS);
    const int n0 = graph_off[g0], n1 = graph_off[g1];
    const int nn = n1 - n0;
    if (threadIdx.x < 2) s_nh[threadIdx.x] = 0;
    __syncthreads();
    // ---- destination side ----
    for (int t = threadIdx.x; t < nn * H; t += 256) {               // light nodes: one thread per (node, head)
        const int v = n0 + t / H, h = t % H;
        const int beg = rowptr_in[v], end = rowptr_in[v + 1];
        if (end - beg > FA_LIGHT) {
            if (h == 0) { const int k = atomicAdd(&s_nh[0], 1); if (k < 256) s_heavy[0][k] = v; }
            continue;
        }
        const float ad = Y[(long long)v * ld_y + F + H + h];
        float al[FA_LIGHT], dl[FA_LIGHT], zs[FA_LIGHT];
#pragma unroll
        for (int i = 0; i < FA_LIGHT; ++i) {                        // clamped, unconditional: all loads of the node go out together
            const int p = min(beg + i, max(end - 1, beg));
            const bool ok = beg + i < end;
            al[i] = ok ? alpha[(long long)p * H + h] : 0.f;
            dl[i] = ok ? dal[(long long)p * H + h] : 0.f;
            zs[i] = ok ? Y[(long long)col_src[p] * ld_y + F + h] : 0.f;
        }
        float S = 0.f, accv = 0.f;
#pragma unroll
        for (int i = 0; i < FA_LIGHT; ++i) S = fmaf(al[i], dl[i], S);
#pragma unroll
        for (int i = 0; i < FA_LIGHT; ++i) {
            const float gz = al[i] * (dl[i] - S) * ((zs[i] + ad > 0.f) ? 1.f : slope);
            if (beg + i < end) dz[(long long)(beg + i) * H + h] = gz;
            accv += (beg + i < end) ? gz : 0.f;
        }
        d_Y[(long long)v * ld_dy + F + H + h] = accv;
    }
    for (int t = threadIdx.x; t < nn * n_pad; t += 256) d_Y[(long long)(n0 + t / n_pad) * ld_dy + F + 2 * H + t % n_pad] = 0.f;
    __syncthreads();
    const bool list_a = s_nh[0] <= 256;                             // (more heavy nodes than the list holds: scan the node range)
    for (int i = w; i < (list_a ? s_nh[0] : nn); i += 4) {          // heavy nodes: one wave each, lanes over the in-edges
        const int v = list_a ? s_heavy[0][i] : n0 + i;
        const int beg = rowptr_in[v], end = rowptr_in[v + 1];
        if (end - beg <= FA_LIGHT) continue;                        // (wave-uniform)
        for (int h = 0; h < H; ++h) {
            const float ad = Y[(long long)v * ld_y + F + H + h];
            float S = 0.f;
            for (int p = beg + l; p < end; p += 64) S = fmaf(alpha[(long long)p * H + h], dal[(long long)p * H + h], S);
            S = wave_sum(S);
            float accv = 0.f;
            for (int p = beg + l; p < end; p += 64) {
                const float de = alpha[(long long)p * H + h] * (dal[(long long)p * H + h] - S);
                const float z = Y[(long long)col_src[p] * ld_y + F + h] + ad;
                const float gz = de * (z > 0.f ? 1.f : slope);
                dz[(long long)p * H + h] = gz;
                accv += gz;
            }
            accv = wave_sum(accv);
            if (l == 0) d_Y[(long long)v * ld_dy + F + H + h] = accv;
        }
    }
    __syncthreads();                                               // dz of this workgroup's edges is complete (first touched below)
    // ---- source side ----
    for (int t = threadIdx.x; t < nn * H; t += 256) {
        const int u = n0 + t / H, h = t % H;
        const int beg = rowptr_out[u], end = rowptr_out[u + 1];
        if (end - beg > FA_LIGHT) {
            if (h == 0) { const int k = atomicAdd(&s_nh[1], 1); if (k < 256) s_heavy[1][k] = u; }
            continue;
        }
        float accu = 0.f;
#pragma unroll
        for (int i = 0; i < FA_LIGHT; ++i) {
            const int j = min(beg + i, max(end - 1, beg));
            accu += (beg + i < end) ? dz[(long long)pos_out[j] * H + h] : 0.f;
        }
        d_Y[(long long)u * ld_dy + F + h] = accu;
    }
    __syncthreads();
    const bool list_b = s_nh[1] <= 256;
    for (int i = w; i < (list_b ? s_nh[1] : nn); i += 4) {
        const int u = list_b ? s_heavy[1][i] : n0 + i;
        const int beg = rowptr_out[u], end = rowptr_out[u + 1];
        if (end - beg <= FA_LIGHT) continue;
        for (int h = 0; h < H; ++h) {
            float accu = 0.f;
            for (int j = beg + l; j < end; j += 64) accu += dz[(long long)pos_out[j] * H + h];
            accu = wave_sum(accu);
            if (l == 0) d_Y[(long long)u * ld_dy + F + h] = accu;
        }
    }
}
// The attention backward of the layer below and stage 1 of the folded layer's reductions depend on the fused sweep only, not on each
// other: one launch, the first nb_attn workgroups do the former.
// ... and (after the egonet-walking sweep) the hubs of graphs cut by its window boundaries: nb_fix = windows - 1 more workgroups.
__global__ __launch_bounds__(256) void gat_attn_bwd_reduce_a_kernel(const AttnBwdArgs aa, const int nb_attn, const TailA a, const HubFixArgs hf,
                                                                    const int nb_fix) {
    // (the fix-up workgroups LAST: almost all of them return after two loads, and in front of the grid they delayed the real jobs by a
    //  dispatch round: 27.9 -> 22.1 us by HIP events)
    const int bid = (int)blockIdx.x, nb_main = (int)gridDim.x - nb_fix;
    if (bid >= nb_main) { fused_hub_fixup_job(bid - nb_main + 1, hf); return; }
    if (bid < nb_attn) { gat_attn_bwd_job(bid, aa); return; }
    reduce_a_job(bid - nb_attn, a);
}

struct CollapseWs {
    float *dZ, *part, *dwa_part, *dwa, *dc, *cn, *dS, *dz, *da1, *da2, *dwv, *ppart, *ppart2;
    void* tail;
    size_t tail_bytes, total;
    int splits, seg_blocks, seg_rows, chunks;
};

// ... and the folded matcher's FORWARD score from the same dot products: <Z_g, Tf[zrow[g]]> = (scale / S_g) sum_{u in g} c~_u e_u -- a sum
// over the graph's few nodes instead of a sweep over Z.
__global__ __launch_bounds__(256) void cl_fold_score_kernel(const int* __restrict__ goff, int G, const float* __restrict__ coef,
                                                            const float* __restrict__ wsum, const float* __restrict__ e_part, int ntile, float scale,
                                                            int apply_exp, float* __restrict__ sc) {
    // one wave per graph: its nodes' tiles are ONE contiguous range of e_part, a lane takes every 64th value (fixed order: deterministic)
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (g >= G) return;
    const int u0 = goff[g], n = (goff[g + 1] - u0) * ntile;
    const float* base = e_part + (long long)u0 * ntile;
    float acc = 0.f;
    for (int i = l; i < n; i += 64) acc = fmaf(coef[u0 + i / ntile], base[i], acc);
    acc = wave_sum(acc);
    if (l == 0) {
        const float S = wsum[g];
        const float raw = S > 0.f ? acc * scale / S : 0.f;
        sc[g] = apply_exp ? __expf(raw) : raw;
    }
}

// phases | 128 of the folded layer's backward entries: the weight-gradient product runs on a second stream BESIDE the caller's dZ product
// and sweeps (every call of one backward pass carries the bit: the workspace layout depends on it).  Few fat k-slices then -- 2 instead
// of the 7 that fill the machine: ~140 workgroups leave the kernels on the caller's stream their wave slots (cl_bwd_dot 73 -> 61 us,
// step -11 us on the 4,096-egonet batch) and the product still ends under the fused sweep (one slice: it does not -- sweep 139 -> 204 us)
constexpr int DW_BESIDE_SPLITS = 2;
static CollapseWs plan_collapse_ws(void* ws, int n, int e, int G, int Kp, int D, int Pd, int vocab, int max_splits = 0) {
    CollapseWs p;
    char* b = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) { float* r = (float*)(b + off); off += align_up(bytes > 0 ? bytes : 4, 256); return r; };
    const int n1 = n > 0 ? n : 1, v1 = vocab > 0 ? vocab : 1;
    p.dZ = take((size_t)(G > 0 ? G : 1) * Kp * 4);
    p.splits = choose_splits(D, Kp, G);
    if (max_splits > 0 && p.splits > max_splits) p.splits = max_splits;
    p.part = take((size_t)p.splits * D * Kp * 4);
    p.chunks = (n + CL_CHUNK - 1) / CL_CHUNK;
    p.dwa_part = take((size_t)(p.chunks > 0 ? p.chunks : 1) * 2 * Kp * 4);
    p.dwa = take((size_t)2 * Kp * 4);
    p.dc = take((size_t)n1 * 4);
    p.cn = take((size_t)n1 * 4);
    p.dS = take((size_t)(G > 0 ? G : 1) * 4);
    p.dz = take((size_t)(e > 0 ? e : 1) * 4);
    p.da1 = take((size_t)n1 * 4);
    p.da2 = take((size_t)n1 * 4);
    p.dwv = take((size_t)n1 * 4);
    p.seg_rows = 64;
    p.seg_blocks = (n + p.seg_rows - 1) / p.seg_rows;
    if (p.seg_blocks < 1) p.seg_blocks = 1;
    p.ppart = take((size_t)p.seg_blocks * v1 * (Pd > 0 ? Pd : 1) * 4);
    p.ppart2 = take((size_t)p.seg_blocks * v1 * 4);
    p.tail_bytes = gemm_tail_ws_bytes();
    p.tail = take(p.tail_bytes);
    p.total = off;
    return p;
}

}  // namespace txe
using namespace txe;
extern "C" {

// extra workspace (behind txe_gat_collapse_ws_bytes) with which txe_gat_collapse_fwd forms hg = Z W^T on the bf16 pipe
static inline size_t collapse_split_bytes(int G, int D, int Kt) {
    const int Kc = round_up(Kt, 16);
    return align_up(split_packed_bytes(G, Kc), 256) + align_up(split_packed_bytes(D, Kc), 256);
}
size_t txe_gat_collapse_split_ws_bytes(int G, int Kh, int Pd, int D) {
    return (G < 1 || Kh < 1 || Pd < 0 || D < 1) ? 0 : collapse_split_bytes(G, D, Kh + Pd);
}
size_t txe_gat_collapse_ws_bytes(int n_nodes, int n_edges, int G, int Kh, int Pd, int D, int vocab) {
    return plan_collapse_ws(nullptr, n_nodes, n_edges, G, round_up(Kh + Pd, 32), D, Pd, vocab).total;
}

// X [N][Kp], Wp [Fp][Kp] (rows < D the weight, rows D / D+1 the folded attention rows), mask: feature-dropout keep bits of X
// or NULL.  pos / pw: WeightedMeanReadout (pw == NULL: MeanReadout).  Saved for backward: a12 [N][2], alpha [E], coef [N],
// wsum [G], gid [N] (graph of each node), Z [G][Kp].  hg [G][D] (row stride ld_hg).
// column tiles per node of txe_gat_collapse_fwd's e_part output; 0 when the batch does not take the chunked Z sweep that forms it
int txe_gat_collapse_e_tiles(int n_nodes, int G, int Kh, int Pd) {
    if (n_nodes <= 0 || G <= 0 || !cl_zsum_chunked(n_nodes, G)) return 0;
    return (round_up(Kh + Pd, 32) / 4 + 63) / 64;
}

// scores of the folded bilinear matcher from txe_gat_collapse_fwd's e_part (the same Tf / zrow): s_g = [exp] <Z_g, Tf[zrow[g]]>
int txe_gat_collapse_fold_scores(const int* graph_off, int n_nodes, int G, int Kh, int Pd, const float* coef, const float* wsum, const float* e_part,
                                 float feat_drop_p, int masked, int apply_exp, float* s, void* stream) {
    if (G < 0 || !graph_off || !coef || !wsum || !e_part || !s || feat_drop_p < 0.f || feat_drop_p >= 1.f) return TXE_ERR_ARG;
    const int nt = txe_gat_collapse_e_tiles(n_nodes, G, Kh, Pd);
    if (nt <= 0) return TXE_ERR_ARG;
    const float fs = (masked && feat_drop_p > 0.f) ? 1.f / (1.f - feat_drop_p) : 1.f;
    ProfScope prof("cl_fold_score_kernel", (hipStream_t)stream, 4.0 * (n_nodes * (nt + 1.0) + 2.0 * G), 1);
    hipLaunchKernelGGL(cl_fold_score_kernel, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, graph_off, G, coef, wsum, e_part, nt, fs, apply_exp, s);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_gat_collapse_fwd(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst, const int* pos_out,
                         const int* graph_off, int n_nodes, int n_edges, int G, const float* X, int Kh, int Pd, const float* Wp, int D,
                         float feat_drop_p, const unsigned* mask, float attn_slope, float attn_drop_p, unsigned long long seed,
                         const int* pos, const float* pw, float* a12, int a12_ready, float* alpha, float* coef, float* wsum, int* gid,
                         float* Z, float* hg, long long ld_hg, const float* Tf, const int* zrow, float* e_part, void* ws, size_t ws_bytes,
                         void* stream) {
    if (n_nodes < 0 || n_edges < 0 || G < 0 || Kh < 1 || Pd < 0 || D < 1 || !rowptr_in || !rowptr_out || !graph_off || !X || !Wp || !a12 ||
        !alpha || !coef || !wsum || !gid || !Z || !ws || (pw && !pos))
        return TXE_ERR_ARG;
    if (feat_drop_p < 0.f || feat_drop_p >= 1.f || attn_drop_p < 0.f || attn_drop_p >= 1.f) return TXE_ERR_ARG;
    const int Kt = Kh + Pd, Kp = round_up(Kt, 32);
    CollapseWs p = plan_collapse_ws(ws, n_nodes, n_edges, G, Kp, D, Pd, 0);
    if (ws_bytes < p.total) return TXE_ERR_WORKSPACE;
    if (G == 0) return TXE_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned* mk = (mask && feat_drop_p > 0.f) ? mask : nullptr;
    const int mask_ld = (Kt + 31) / 32;
    const float fs = mk ? 1.f / (1.f - feat_drop_p) : 1.f, as = 1.f / (1.f - attn_drop_p);
    const float* wa = Wp + (long long)D * Kp;
    const unsigned* dummy_mask = reinterpret_cast<const unsigned*>(X);     // never dereferenced by the <false> instantiations
    if (n_nodes > 0) {
        const int nb = (n_nodes + 3) / 4;
        if (!(a12_ready & 1)) {    // (the producer of X may already have formed them: txe_gat_aggregate_fwd's fused epilogue)
            ProfScope prof(mk ? "cl_logits_kernel<true>" : "cl_logits_kernel<false>", s, 4.0 * n_nodes * (double)Kp, 1);
            if (mk) hipLaunchKernelGGL(cl_logits_kernel<true>, dim3(nb < 2048 ? nb : 2048), dim3(256), 0, s, X, Kp, n_nodes, mk, mask_ld, fs, wa, a12);
            else hipLaunchKernelGGL(cl_logits_kernel<false>, dim3(nb < 2048 ? nb : 2048), dim3(256), 0, s, X, Kp, n_nodes, dummy_mask, mask_ld, fs, wa, a12);
        }
        hipLaunchKernelGGL(cl_attn_coef_kernel, dim3((G + CG_GRAPHS - 1) / CG_GRAPHS), dim3(256), 0, s, rowptr_in, col_src, rowptr_out, col_dst, pos_out,
                           graph_off, G, (const float*)a12, attn_slope, attn_drop_p, as, seed, pos, pw, alpha, coef, wsum, gid);
        TXE_CHECK_LAUNCH();
    } else if (G > 0) {
        hipLaunchKernelGGL(cl_wsum_kernel, dim3((G + 3) / 4), dim3(256), 0, s, graph_off, G, pos, pw, wsum, gid);
    }
    // e_part != NULL (with hg == NULL: the folded matcher already has Tf [runs][Kp] and zrow [G], graph -> its row of Tf): the sweep also
    // leaves <Tf[zrow[g]], keep X[u]> per node and column tile at e_part [N][txe_gat_collapse_e_tiles] -- backward's <dZ, X> sweep, ahead of time
    const int rc_z = cl_zsum_launch(graph_off, G, n_nodes, X, Kp, mk, dummy_mask, mask_ld, fs, (const float*)coef, (const float*)wsum, Z, s, Tf, zrow,
                                    e_part);
    if (rc_z) return rc_z;
    if (!hg) return TXE_OK;          // (the caller folds hg = Z W^T into what consumes it: txe_bilinear_folded_*)
    if (G > 0 && (a12_ready & 2)) {
        if (ws_bytes < p.total + collapse_split_bytes(G, D, Kt)) return TXE_ERR_WORKSPACE;            // (the route is the caller's choice, not the buffer's size)
        // hg = Z W^T on the bf16 matrix pipe (txe_gemm_split.h): Z and the weight rows packed behind the workspace's own regions
        char* sw = (char*)ws + p.total;
        const int Kc = round_up(Kt, 16);
        const size_t ba = align_up(split_packed_bytes(G, Kc), 256);
        int rc = split_pack_launch(Z, Kp, G, Kc, 0, sw, s);
        if (rc) return rc;
        rc = split_pack_launch(Wp, Kp, D, Kc, 1, sw + ba, s);
        if (rc) return rc;
        return gemm_nt_split_launch(sw, sw + ba, G, D, Kc, hg, ld_hg, 2.0 * G * (double)D * Kt, s);
    }
    VMat A = vmat_plain(Z, Kp, G, Kp);
    VMat B = vmat_plain(Wp, Kp, round_up(D + 2, 128), Kp);       // all Fp packed rows are readable: every tile stays on the plain loader
    Epi E = epi_plain(hg, ld_hg, D);
    E.alg_flops = 2.0 * G * (double)D * Kt;
    return gemm_nt(A, B, E, G, D, Kp, 1, s, p.tail, p.tail_bytes);
}

// d_hg [G][D] -> d_X [N][Kp] (first Kh columns through leaky' of X when act_on: they are d(pre-activation) of the previous
// layer), dW [D][Kt], d_attn_l / d_attn_r [D], dP [vocab][Pd] (Pd > 0), d_pw [vocab] (pw != NULL).
int txe_gat_collapse_bwd(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst, const int* pos_out,
                         const int* graph_off, int n_nodes, int n_edges, int G, const float* X, int Kh, int Pd, const int* pos, int vocab,
                         const float* Wp, const float* W, const float* attn_l, const float* attn_r, int D, float feat_drop_p,
                         const unsigned* mask, float attn_slope, float attn_drop_p, unsigned long long seed, const float* pw,
                         const float* a12, const float* alpha, const float* coef, const float* wsum, const int* gid, const float* Z,
                         const float* hg, long long ld_hg, const float* d_hg, long long ld_dhg, int act_on, float act_slope, float* d_X, float* dW, float* d_attn_l,
                         float* d_attn_r, float* dP, float* d_pw, void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || n_edges < 0 || G < 0 || Kh < 1 || Pd < 0 || D < 1 || !rowptr_in || !rowptr_out || !graph_off || !X || !Wp || !W ||
        !attn_l || !attn_r || !a12 || !alpha || !coef || !wsum || !gid || !Z || !hg || !d_hg || !d_X || !dW || !d_attn_l || !d_attn_r || !ws)
        return TXE_ERR_ARG;
    if ((Pd > 0 || pw) && (!pos || vocab < 1 || vocab > MAX_VOCAB)) return TXE_ERR_ARG;
    if ((Pd > 0 && !dP) || (pw && !d_pw)) return TXE_ERR_ARG;
    if (feat_drop_p < 0.f || feat_drop_p >= 1.f || attn_drop_p < 0.f || attn_drop_p >= 1.f) return TXE_ERR_ARG;
    const int Kt = Kh + Pd, Kp = round_up(Kt, 32);
    CollapseWs p = plan_collapse_ws(ws, n_nodes, n_edges, G, Kp, D, Pd, vocab);
    if (ws_bytes < p.total) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const unsigned* mk = (mask && feat_drop_p > 0.f) ? mask : nullptr;
    const int mask_ld = (Kt + 31) / 32;
    const float fs = mk ? 1.f / (1.f - feat_drop_p) : 1.f, as = 1.f / (1.f - attn_drop_p);
    const float* wa = Wp + (long long)D * Kp;
    const unsigned* dummy_mask = reinterpret_cast<const unsigned*>(X);
    int rc;
    // ---- dZ = d_hg W ;  dW (main part, split-K partial slices) = d_hg^T Z ----
    {
        VMat A = vmat_plain(d_hg, ld_dhg, G, D);
        VMat B = vmat_plain(Wp, Kp, D, Kp);
        Epi E = epi_plain(p.dZ, Kp, Kp);
        E.alg_flops = 2.0 * G * (double)Kt * D;
        rc = gemm_nn(A, B, E, G, Kp, D, 1, s, p.tail, p.tail_bytes);
        if (rc) return rc;
    }
    const long long split_stride = (long long)D * Kp;
    {
        VMat A = vmat_plain(d_hg, ld_dhg, G, D);
        VMat B = vmat_plain(Z, Kp, G, Kp);
        Epi E = epi_plain(p.part, Kp, Kp);
        E.split_stride = split_stride;
        E.alg_flops = 2.0 * D * (double)Kt * G;
        rc = gemm_tn(A, B, E, D, Kp, G, p.splits, s);
        if (rc) return rc;
    }
    const int S = G > 0 ? p.splits : 0;
    const int nblk = (G > 0 && n_nodes > 0) ? p.chunks : 0;
    if (G > 0 && n_nodes > 0) {
        const int ntile = (Kp / 4 + 63) / 64;
        rc = cl_bwd_dot_launch(n_nodes, gid, X, Kp, mk, dummy_mask, mask_ld, fs, (const float*)p.dZ, wsum, coef, p.dc, p.cn, (G + 3) / 4, G, D, d_hg, ld_dhg,
                               hg, ld_hg, p.dS, 4.0 * ((n_nodes + (double)G) * Kp + 2.0 * G * D), s);
        if (rc) return rc;
        hipLaunchKernelGGL(cl_attn_bwd_kernel<false>, dim3((G + CG_GRAPHS - 1) / CG_GRAPHS), dim3(256), 0, s, rowptr_in, col_src, rowptr_out, pos_out,
                           graph_off, G, a12, attn_slope, alpha, attn_drop_p, as, seed, pos, pw, (const float*)p.dc, (const float*)p.dS, p.dz,
                           p.da1, p.da2, p.dwv, FoldDcArgs{});
        {
            const long long nwaves = (long long)p.chunks * ntile;
            ProfScope prof(mk ? "cl_bwd_dx_kernel<true, true>" : "cl_bwd_dx_kernel<false, true>", s, 4.0 * (2.0 * n_nodes + G) * Kp, 1);
            if (mk) hipLaunchKernelGGL((cl_bwd_dx_kernel<true, true>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, s, n_nodes, ntile, gid, X, Kp, Kh, mk,
                                       mask_ld, fs, (const float*)p.dZ, (const float*)p.cn, (const float*)p.da1, (const float*)p.da2, wa, act_on,
                                       act_slope, d_X, p.dwa_part);
            else hipLaunchKernelGGL((cl_bwd_dx_kernel<false, true>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, s, n_nodes, ntile, gid, X, Kp, Kh,
                                    dummy_mask, mask_ld, fs, (const float*)p.dZ, (const float*)p.cn, (const float*)p.da1, (const float*)p.da2, wa,
                                    act_on, act_slope, d_X, p.dwa_part);
        }
        TXE_CHECK_LAUNCH();
    }
    // ---- phase A: d_wa = sum of the per-block partials; partial position sums (embedding / readout position-weight gradients) ----
    const int nseg = n_nodes > 0 ? p.seg_blocks : 0;
    TailA ta;
    memset(&ta, 0, sizeof(ta));
    ta.nb_s1a = Pd > 0 ? nseg : 0; ta.s1a = Seg1Args{d_X + Kh, (long long)Kp, Pd, p.ppart};
    ta.nb_s1b = pw ? nseg : 0; ta.s1b = Seg1Args{p.dwv, 1, 1, p.ppart2};
    ta.pos = pos; ta.n_rows = n_nodes; ta.vocab = vocab; ta.rows_per_block = p.seg_rows;
    ta.r_kind = 2; ta.nb_r = (2 * Kp + 63) / 64; ta.r2 = Seg2Args{p.dwa_part, nblk, 2 * Kp, p.dwa};
    hipLaunchKernelGGL(gat_bwd_reduce_a_kernel, dim3(ta.nb_s1a + ta.nb_s1b + ta.nb_r), dim3(256), 0, s, ta);
    TXE_CHECK_LAUNCH();
    // ---- phase B: dW = main + attn (x) d_wa, d_attn = <d_wa, W> (unfold);  dP, d_pw ----
    TailB tb;
    memset(&tb, 0, sizeof(tb));
    tb.nb_u = D;
    tb.u = UnfoldArgs{p.part, S, split_stride, p.dwa, (long long)Kp, W, (long long)Kt, attn_l, attn_r, 1, D, Kt, dW, (long long)Kt, d_attn_l,
                      d_attn_r};
    tb.nb_2a = Pd > 0 ? (vocab * Pd + 63) / 64 : 0;
    tb.s2a = Seg2Args{p.ppart, nseg, vocab * Pd, dP};
    tb.nb_2b = pw ? (vocab + 63) / 64 : 0;
    tb.s2b = Seg2Args{p.ppart2, nseg, vocab, d_pw};
    hipLaunchKernelGGL(gat_bwd_reduce_b_kernel, dim3(tb.nb_u + tb.nb_2a + tb.nb_2b), dim3(256), 0, s, tb);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"

namespace txe {
struct FusedWs {
    CollapseWs c;
    float *dal, *dwa_part, *ppart, *hpart;
    int nblocks, npw;
    size_t total;
};
static FusedWs plan_fused_ws(void* ws, int n, int e, int G, int Kh, int Kp, int D, int Pd, int vocab, int Hp, int max_splits = 0) {
    FusedWs f;
    f.c = plan_collapse_ws(ws, n, e, G, Kp, D, Pd, vocab, max_splits);
    char* b = (char*)ws;
    size_t off = f.c.total;
    auto take = [&](size_t bytes) { float* r = (float*)(b + off); off += align_up(bytes > 0 ? bytes : 4, 256); return r; };
    f.npw = fb_nodes_per_wg(n, (Kh > 2048) ? 2 : 3);                // (rows of more than 2,048 feature columns: NI >= 3, two workgroups per CU)
    f.nblocks = (n + f.npw - 1) / f.npw;
    const int nb1 = f.nblocks > 0 ? f.nblocks : 1;
    f.dal = take((size_t)(e > 0 ? e : 1) * Hp * 4);
    f.dwa_part = take((size_t)nb1 * 2 * Kp * 4);
    f.ppart = take((size_t)nb1 * (vocab > 0 ? vocab : 1) * (Pd > 0 ? Pd : 1) * 4);
    f.hpart = take(Hp == 4 ? (size_t)nb1 * Kp * 4 : 4);             // (the egonet walk: H*D = Kh <= Kp floats per window)
    f.total = off;
    return f;
}
}  // namespace txe

extern "C" {

// The walk plan of a batch of graphs for the egonet-walking sweeps (egonet_walk_plan_kernel): 8 ints per node.
size_t txe_egonet_walk_plan_bytes(int n_nodes) { return (size_t)(n_nodes > 0 ? n_nodes : 1) * EGO_PLAN_W * sizeof(int); }
int txe_egonet_walk_plan(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst, const int* pos_out,
                         const int* graph_off, int n_nodes, int G, int* plan, void* stream) {
    if (n_nodes < 0 || G < 0 || !plan || (n_nodes > 0 && (!rowptr_in || !col_src || !rowptr_out || !col_dst || !pos_out || !graph_off))) return TXE_ERR_ARG;
    if (((uintptr_t)plan & 15) != 0) return TXE_ERR_ARG;
    if (n_nodes == 0 || G == 0) return TXE_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("egonet_walk_plan_kernel", s, 4.0 * (6.0 * n_nodes + EGO_PLAN_W * (double)n_nodes), 1);
    hipLaunchKernelGGL(egonet_walk_plan_kernel, dim3((G + 3) / 4), dim3(256), 0, s, rowptr_in, col_src, rowptr_out, col_dst, pos_out, graph_off, G, plan);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// 1 when txe_gat_collapse_bwd_fused supports the shape: the previous layer has 1, 2 or 4 heads, its H*D columns are a multiple of 16
// and at most 4096, and the folded layer's input has at most 128 columns behind them.
int txe_gat_fused_bwd_supported(int Kh, int Pd, int Hp, int Dp) {
    const int F = Hp * Dp, Kp = round_up(Kh + Pd, 32);
    return (Hp == 1 || Hp == 2 || Hp == 4) && F == Kh && (F % 16) == 0 && F <= 4096 && Kp - F <= FB_MAXPD && Pd <= FB_MAXPD && (Dp % 4) == 0;
}

size_t txe_gat_collapse_bwd_fused_ws_bytes(int n_nodes, int n_edges, int G, int Kh, int Pd, int D, int vocab, int Hp) {
    return plan_fused_ws(nullptr, n_nodes, n_edges, G, Kh, round_up(Kh + Pd, 32), D, Pd, vocab, Hp).total;
}

// txe_gat_collapse_bwd FUSED with txe_gat_aggregate_bwd of the layer below: same inputs as txe_gat_collapse_bwd plus
// that layer's projection output Yp [N][ld_yp] = [ft | a1 | a2] (Hp heads of Dp columns, Hp*Dp == Kh), its attention alpha_p [E][Hp]
// (destination-CSR order), attention slope / dropout / seed.  Instead of d_X it returns that layer's d_Yp [N][ld_dyp] =
// [d_ft | d_a1 | d_a2 | n_pad zero columns] directly; dz_p [E][Hp] is scratch.  act_slope: slope of the activation between the two
// layers (1 = none).  dP / d_pw / dW / d_attn as txe_gat_collapse_bwd.  phases: 15 = everything; or, for a caller that overlaps the
// independent weight-gradient GEMM with the sweeps on a second stream, separate calls with 1 (dZ GEMM), 2 (dW GEMM partials: needs
// only d_hg and Z), 4 (sweeps + first reduction stage: needs 1), 8 (final reductions: needs 2 and 4) and the same workspace.
// phases | 1024: the source-side sweep does not walk egonets from registers (gat_fused_bwd_kernel for every head count: the A/B switch).
int txe_gat_collapse_bwd_fused(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst, const int* pos_out,
                               const int* graph_off, int n_nodes, int n_edges, int G, const float* X, int Kh, int Pd, const int* pos,
                               int vocab, const float* Wp, const float* W, const float* attn_l, const float* attn_r, int D,
                               float feat_drop_p, const unsigned* mask, float attn_slope, float attn_drop_p, unsigned long long seed,
                               const float* pw, const float* a12, const float* alpha, const float* coef, const float* wsum,
                               const int* gid, const float* Z, const float* hg, long long ld_hg, const float* d_hg, long long ld_dhg,
                               float act_slope, const float* Yp, long long ld_yp, int Hp, int Dp, float attn_slope_p,
                               float attn_drop_p_p, unsigned long long seed_p, const float* alpha_p, float* d_Yp, long long ld_dyp,
                               int n_pad, float* dz_p, float* dW, float* d_attn_l, float* d_attn_r, float* dP, float* d_pw, int phases,
                               const float* dw_main, int dw_slices, const float* e_part, const float* m_ds, const float* m_s, int m_exp,
                               const float* Tf, const int* zrow, int* zgid, const int* walk_plan, void* chain, void* ws, size_t ws_bytes,
                               void* stream) {
    // phases | 512 (with | 256): the <dZ, X> sweep was done in forward (txe_gat_collapse_fwd's e_part); m_ds / m_s [G]: the folded matcher's
    // score gradient and scores, m_exp: it exponentiates -- see cl_fold_dc_kernel
    // phases | 256: `d_hg` IS dZ [G][Kp] (ld_dhg its row pitch) -- whoever consumed Z folded hg = Z W^T into its own product
    // (txe_bilinear_folded_*) and hands back dZ and the main part of dW as dw_slices slices [D][Kp] at dw_main (summed in order; 0: none)
    const bool dz_given = (phases & 256) != 0;
    if (n_nodes < 0 || n_edges < 0 || G < 0 || Kh < 1 || Pd < 0 || D < 1 || !rowptr_in || !rowptr_out || !graph_off || !X || !Wp || !W ||
        !attn_l || !attn_r || !a12 || !alpha || !coef || !wsum || !gid || !Z || (!hg && !dz_given) || (!d_hg && !(phases & 512)) || !dW || !d_attn_l || !d_attn_r ||
        !ws || !Yp || !alpha_p || !d_Yp || !dz_p || n_pad < 0 || dw_slices < 0 || (dw_slices > 0 && !dw_main))
        return TXE_ERR_ARG;
    if (!txe_gat_fused_bwd_supported(Kh, Pd, Hp, Dp)) return TXE_ERR_ARG;
    if ((Pd > 0 || pw) && (!pos || vocab < 1 || vocab > MAX_VOCAB)) return TXE_ERR_ARG;
    if ((Pd > 0 && !dP) || (pw && !d_pw)) return TXE_ERR_ARG;
    if (feat_drop_p < 0.f || feat_drop_p >= 1.f || attn_drop_p < 0.f || attn_drop_p >= 1.f || attn_drop_p_p < 0.f || attn_drop_p_p >= 1.f)
        return TXE_ERR_ARG;
    const int Kt = Kh + Pd, Kp = round_up(Kt, 32), F = Hp * Dp;
    FusedWs fw = plan_fused_ws(ws, n_nodes, n_edges, G, Kh, Kp, D, Pd, vocab, Hp, (phases & 128) ? DW_BESIDE_SPLITS : 0);
    CollapseWs& p = fw.c;
    if (ws_bytes < fw.total) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const unsigned* mk = (mask && feat_drop_p > 0.f) ? mask : nullptr;
    const int mask_ld = (Kt + 31) / 32;
    const float fs = mk ? 1.f / (1.f - feat_drop_p) : 1.f, as = 1.f / (1.f - attn_drop_p);
    const float* wa = Wp + (long long)D * Kp;
    const unsigned* dummy_mask = reinterpret_cast<const unsigned*>(X);
    int rc;
    if (dz_given) phases &= ~3;
    const float* const dZv = dz_given ? d_hg : (const float*)p.dZ;
    const long long ld_dz = dz_given ? ld_dhg : (long long)Kp;
    if (dz_given && ld_dz != Kp) return TXE_ERR_ARG;               // (the sweeps walk dZ rows with the padded pitch)
    if (phases & 1) {   // dZ = d_hg W
        VMat A = vmat_plain(d_hg, ld_dhg, G, D);
        VMat B = vmat_plain(Wp, Kp, D, Kp);
        Epi E = epi_plain(p.dZ, Kp, Kp);
        E.alg_flops = 2.0 * G * (double)Kt * D;
        rc = gemm_nn(A, B, E, G, Kp, D, 1, s, p.tail, p.tail_bytes);
        if (rc) return rc;
    }
    const long long split_stride = (long long)D * Kp;
    if (phases & 2) {   // dW (main part, split-K partial slices) = d_hg^T Z
        VMat A = vmat_plain(d_hg, ld_dhg, G, D);
        VMat B = vmat_plain(Z, Kp, G, Kp);
        Epi E = epi_plain(p.part, Kp, Kp);
        E.split_stride = split_stride;
        E.alg_flops = 2.0 * D * (double)Kt * G;
        rc = gemm_tn(A, B, E, D, Kp, G, p.splits, s);
        if (rc) return rc;
    }
    const int S = dz_given ? dw_slices : (G > 0 ? p.splits : 0);
    const float* const partv = dz_given ? dw_main : (const float*)p.part;
    const int nblk = (G > 0 && n_nodes > 0) ? fw.nblocks : 0;
    if ((phases & 4) && G > 0 && n_nodes > 0) {

        FoldDcArgs fdc{};
        if (phases & 512) {
            if (!dz_given || !e_part || !m_ds || !m_s || !Tf || !zrow || !zgid) return TXE_ERR_ARG;
            const int nt_e = txe_gat_collapse_e_tiles(n_nodes, G, Kh, Pd);
            if (nt_e <= 0) return TXE_ERR_ARG;
            fdc = FoldDcArgs{e_part, nt_e, m_ds, m_s, m_exp, fs, wsum, coef, p.dc, p.cn, p.dS, zrow, zgid};      // (the edge kernel's prologue)
        } else {
        // (dS[g] = -<dZ[g], Z[g]> / S_g; with d_hg at hand it is <d_hg[g], hg[g]>, D columns instead of Kp)
        rc = cl_bwd_dot_launch(n_nodes, gid, X, Kp, mk, dummy_mask, mask_ld, fs, dZv, wsum, coef, p.dc, p.cn, (G + 3) / 4, G, dz_given ? Kp : D,
                               d_hg, ld_dhg, dz_given ? Z : hg, dz_given ? (long long)Kp : ld_hg, p.dS, 4.0 * ((n_nodes + (double)G) * Kp + 2.0 * G * D), s);
        if (rc) return rc;
        }
        if (phases & 512)
            hipLaunchKernelGGL(cl_attn_bwd_kernel<true>, dim3((G + CG_GRAPHS - 1) / CG_GRAPHS), dim3(256), 0, s, rowptr_in, col_src, rowptr_out, pos_out,
                               graph_off, G, a12, attn_slope, alpha, attn_drop_p, as, seed, pos, pw, (const float*)p.dc, (const float*)p.dS, p.dz,
                               p.da1, p.da2, p.dwv, fdc);
        else
            hipLaunchKernelGGL(cl_attn_bwd_kernel<false>, dim3((G + CG_GRAPHS - 1) / CG_GRAPHS), dim3(256), 0, s, rowptr_in, col_src, rowptr_out, pos_out,
                               graph_off, G, a12, attn_slope, alpha, attn_drop_p, as, seed, pos, pw, (const float*)p.dc, (const float*)p.dS, p.dz,
                               p.da1, p.da2, p.dwv, fdc);
        {
            FusedBwdArgs a;
            memset(&a, 0, sizeof(a));
            a.rowptr_out = rowptr_out; a.col_dst = col_dst; a.pos_out = pos_out; a.gid = (phases & 512) ? (const int*)zgid : gid; a.pos = pos ? pos : gid;
            a.n_nodes = n_nodes;
            a.X = X; a.Kp = Kp; a.Kh = Kh; a.Pd = Pd; a.mask = mk ? mk : dummy_mask; a.mask_ld = mask_ld; a.fscale = fs;
            a.dZ = (phases & 512) ? Tf : dZv; a.cn = p.cn; a.da1 = p.da1; a.da2 = p.da2; a.wa = wa; a.act_slope = act_slope; a.vocab = vocab > 0 ? vocab : 1;
            a.Y = Yp; a.ld_y = ld_yp; a.H = Hp; a.D = Dp; a.alpha = alpha_p; a.drop_p = attn_drop_p_p;
            a.drop_scale = 1.f / (1.f - attn_drop_p_p); a.seed = seed_p;
            a.d_Y = d_Yp; a.ld_dy = ld_dyp; a.dal = fw.dal; a.dwa_part = fw.dwa_part; a.ppart = fw.ppart;
            a.npw = fw.npw;
            a.rowptr_in = rowptr_in; a.col_src = col_src; a.goff = graph_off; a.ggid = gid; a.G = G; a.hpart = fw.hpart;
            a.plan = walk_plan;
            const int nvec = F / 16, ni = (nvec + 63) / 64, nwh = 4 / Hp;
            // algorithmic bytes: read X' (own row + once per out-edge is an L2 matter), dZ, Y; write d_Y
            char name[64];
            const bool ego = Hp == 4 && !(phases & 1024);            // one head per wave: the egonet-walking variant (generic graphs inside)
            if (ego) snprintf(name, sizeof(name), "gat_fused_bwd_ego_kernel<%s, %d>", mk ? "true" : "false", ni);
            else snprintf(name, sizeof(name), "gat_fused_bwd_kernel<%s, %d, %d>", mk ? "true" : "false", ni, nwh);
            ProfScope prof(name, s, 4.0 * (n_nodes * ((double)Kp + 2.0 * F) + (double)G * Kp), 1);
#define TXE_FB(M_, NI_, NW_) hipLaunchKernelGGL((gat_fused_bwd_kernel<M_, NI_, NW_>), dim3(fw.nblocks), dim3(256), (size_t)(4 * Kp + a.vocab * (Pd > 0 ? Pd : 1)) * sizeof(float), s, a)
#define TXE_FB_NI(M_, NW_) do { if (ni == 1) TXE_FB(M_, 1, NW_); else if (ni == 2) TXE_FB(M_, 2, NW_); else if (ni == 3) TXE_FB(M_, 3, NW_); else TXE_FB(M_, 4, NW_); } while (0)
#define TXE_FB_NW(M_) do { if (nwh == 1) TXE_FB_NI(M_, 1); else if (nwh == 2) TXE_FB_NI(M_, 2); else TXE_FB_NI(M_, 4); } while (0)
            if (ego) {
#define TXE_FBE(M_, NI_) hipLaunchKernelGGL((gat_fused_bwd_ego_kernel<M_, NI_>), dim3(fw.nblocks), dim3(256), (size_t)(4 * Kp + a.vocab * (Pd > 0 ? Pd : 1)) * sizeof(float), s, a)
#define TXE_FBE_NI(M_) do { if (ni == 1) TXE_FBE(M_, 1); else if (ni == 2) TXE_FBE(M_, 2); else if (ni == 3) TXE_FBE(M_, 3); else TXE_FBE(M_, 4); } while (0)
                if (mk) TXE_FBE_NI(true); else TXE_FBE_NI(false);
#undef TXE_FBE_NI
#undef TXE_FBE
            } else if (mk) TXE_FB_NW(true); else TXE_FB_NW(false);
#undef TXE_FB_NW
#undef TXE_FB_NI
#undef TXE_FB
        }
        TXE_CHECK_LAUNCH();
    }
    // ---- the layer below's attention backward (edge level, from the sweep's raw d alpha) + phase A: d_wa = sum of the per-workgroup
    //      partials; readout position-weight partial sums -- one launch ----
    const int nseg = n_nodes > 0 ? p.seg_blocks : 0;
    if (phases & 4) {
    TailA ta;
    memset(&ta, 0, sizeof(ta));
    ta.nb_s1a = 0;
    ta.nb_s1b = pw ? nseg : 0; ta.s1b = Seg1Args{p.dwv, 1, 1, p.ppart2};
    ta.pos = pos; ta.n_rows = n_nodes; ta.vocab = vocab; ta.rows_per_block = p.seg_rows;
    ta.r_kind = 2; ta.nb_r = (2 * Kp + 63) / 64; ta.r2 = Seg2Args{fw.dwa_part, nblk, 2 * Kp, p.dwa};
    const bool attn = G > 0 && n_nodes > 0;
    AttnBwdArgs aa{rowptr_in, col_src, rowptr_out, pos_out, graph_off, G, Yp, ld_yp, Hp, F, attn_slope_p, alpha_p, (const float*)fw.dal, dz_p, d_Yp,
                   ld_dyp, n_pad};
    const int nb_attn = attn ? (G + FA_GRAPHS - 1) / FA_GRAPHS : 0;
    const bool ego = Hp == 4 && !(phases & 1024) && attn;
    HubFixArgs hf{graph_off, gid, rowptr_out, col_dst, n_nodes, fw.npw, fw.nblocks, F, fw.hpart, d_Yp, ld_dyp};
    const int nb_fix = ego ? fw.nblocks - 1 : 0;
    ProfScope prof("gat_attn_bwd_reduce_a_kernel", s, attn ? 4.0 * (n_edges * (4.0 * Hp + 2.0) + n_nodes * (4.0 * Hp + n_pad)) : 0.0, 1);
    hipLaunchKernelGGL(gat_attn_bwd_reduce_a_kernel, dim3(nb_fix + nb_attn + ta.nb_s1b + ta.nb_r), dim3(256), 0, s, aa, nb_attn, ta, hf, nb_fix);
    TXE_CHECK_LAUNCH();
    }
    if (!(phases & 8)) return TXE_OK;
    // ---- phase B: dW = main + attn (x) d_wa, d_attn = <d_wa, W> (unfold);  dP (from the fused sweep's partials), d_pw ----
    TailB tb;
    memset(&tb, 0, sizeof(tb));
    tb.nb_u = D;
    tb.u = UnfoldArgs{partv, S, split_stride, p.dwa, (long long)Kp, W, (long long)Kt, attn_l, attn_r, 1, D, Kt, dW, (long long)Kt, d_attn_l,
                      d_attn_r};
    tb.nb_2a = Pd > 0 ? (vocab * Pd + 63) / 64 : 0;
    tb.s2a = Seg2Args{fw.ppart, nblk, vocab * Pd, dP};
    tb.nb_2b = pw ? (vocab + 63) / 64 : 0;
    tb.s2b = Seg2Args{p.ppart2, nseg, vocab, d_pw};
    return tail_b_submit(&tb, chain, (phases & 64) != 0, s);
}

}  // extern "C"

// =====================================================================================================================
// Last GCNLayer folded behind MeanReadout / WeightedMeanReadout (PGCN / GCN output layer: no activation; model_zoo.py:35-47,
// 139-167, 227-242):  hg[g] = sum_v w_v/S_g (norm_v sum_{u->v} norm_u Xd[u] W + b) = (sum_{u in g} c_u Xd[u]) W + b,
//     c_u = norm_u sum_{v : u->v} w_v norm_v / S_g      -- graph constants (no attention): one sweep forward, one backward
// (two with learnable readout weights).  Reuses the sweep kernels of the GAT fold above.
// =====================================================================================================================
namespace txe {

// one wave per source: c~_u = norm_u * sum_{j in out(u)} w_{dst(j)} norm_{dst(j)}
__global__ __launch_bounds__(256) void gcl_coef_kernel(const int* __restrict__ rowptr_out, const int* __restrict__ col_dst, int n_nodes,
                                                       const float* __restrict__ norm, const int* __restrict__ pos,
                                                       const float* __restrict__ pw, float* __restrict__ coef) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + w;
    if (u >= n_nodes) return;
    float c = 0.f;
    for (int j = rowptr_out[u] + l; j < rowptr_out[u + 1]; j += 64) {
        const int v = col_dst[j];
        c = fmaf(pw ? cl_softplus(pw[pos[v]]) : 1.f, norm[v], c);
    }
    c = wave_sum(c);
    if (l == 0) coef[u] = c * norm[u];
}

// one wave per destination: dwv[v] = (dS_g(v) + norm_v sum_{p in in(v)} norm_u dc~_u) * sigmoid(pw[pos_v])
__global__ __launch_bounds__(256) void gcl_bwd_w_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int n_nodes,
                                                        const float* __restrict__ norm, const int* __restrict__ pos,
                                                        const float* __restrict__ pw, const float* __restrict__ dc,
                                                        const float* __restrict__ dS, const int* __restrict__ gid,
                                                        float* __restrict__ dwv) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + w;
    if (v >= n_nodes) return;
    float a = 0.f;
    for (int p = rowptr[v] + l; p < rowptr[v + 1]; p += 64) a = fmaf(norm[col[p]], dc[col[p]], a);
    a = wave_sum(a);
    if (l == 0) dwv[v] = (dS[gid[v]] + norm[v] * a) * cl_sigmoid(pw[pos[v]]);
}

// y[g][f] += b[f]
__global__ void gcl_add_bias_kernel(float* __restrict__ y, long long ld, int rows, int cols, const float* __restrict__ b) {
    const long long n = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[(i / cols) * ld + (i % cols)] += b[i % cols];
}

struct GclWs {
    float *dZ, *part, *dc, *cn, *dS, *dwv, *ppart, *ppart2, *cpart;
    void* tail;
    size_t tail_bytes, total;
    int splits, seg_blocks, seg_rows;
};

static GclWs plan_gcl_ws(void* ws, int n, int G, int Kp, int Fop, int Pd, int vocab) {
    GclWs p;
    char* b = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) { float* r = (float*)(b + off); off += align_up(bytes > 0 ? bytes : 4, 256); return r; };
    const int n1 = n > 0 ? n : 1, g1 = G > 0 ? G : 1, v1 = vocab > 0 ? vocab : 1;
    p.dZ = take((size_t)g1 * Kp * 4);
    p.splits = choose_splits(Kp, Fop, G);
    p.part = take((size_t)p.splits * Kp * Fop * 4);
    p.dc = take((size_t)n1 * 4);
    p.cn = take((size_t)n1 * 4);
    p.dS = take((size_t)g1 * 4);
    p.dwv = take((size_t)n1 * 4);
    p.seg_rows = 64;
    p.seg_blocks = (n + p.seg_rows - 1) / p.seg_rows;
    if (p.seg_blocks < 1) p.seg_blocks = 1;
    p.ppart = take((size_t)p.seg_blocks * v1 * (Pd > 0 ? Pd : 1) * 4);
    p.ppart2 = take((size_t)p.seg_blocks * v1 * 4);
    p.cpart = take(colsum_ws_bytes(G, Fop));
    p.tail_bytes = gemm_tail_ws_bytes();
    p.tail = take(p.tail_bytes);
    p.total = off;
    return p;
}

// cn[u] = coef[u] / S_g(u)   (MeanReadout path: no dot sweep to piggy-back on)
__global__ void gcl_cn_kernel(int n_nodes, const int* __restrict__ gid, const float* __restrict__ coef, const float* __restrict__ wsum,
                              float* __restrict__ cn) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_nodes) return;
    const float S = wsum[gid[u]];
    cn[u] = S > 0.f ? coef[u] / S : 0.f;
}

}  // namespace txe
using namespace txe;
extern "C" {

size_t txe_gcn_collapse_ws_bytes(int n_nodes, int G, int Kh, int Pd, int Fo, int vocab) {
    return plan_gcl_ws(nullptr, n_nodes, G, round_up(Kh + Pd, 32), round_up(Fo, 32), Pd, vocab).total;
}

// X [N][Kp], Wp [Kp128][Fop], mask as for txe_gcn_dense_*; norm [N] (txe_gcn_norm); bias [Fo] or NULL; pw == NULL: MeanReadout.
// Saved for backward: coef [N], wsum [G], gid [N], Z [G][Kp].  hg [G][Fo] (row stride ld_hg).
int txe_gcn_collapse_fwd(const int* rowptr_out, const int* col_dst, const int* graph_off, int n_nodes, int G, const float* X, int Kh, int Pd,
                         const float* Wp, int Fo, const float* bias, float drop_p, const unsigned* mask, const float* norm, const int* pos,
                         const float* pw, float* coef, float* wsum, int* gid, float* Z, float* hg, long long ld_hg, void* ws,
                         size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || G < 0 || Kh < 1 || Pd < 0 || Fo < 1 || !rowptr_out || !graph_off || !X || !Wp || !norm || !coef || !wsum || !gid || !Z ||
        !ws || (pw && !pos))
        return TXE_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return TXE_ERR_ARG;
    const int Kt = Kh + Pd, Kp = round_up(Kt, 32), Fop = round_up(Fo, 32);
    GclWs p = plan_gcl_ws(ws, n_nodes, G, Kp, Fop, Pd, 0);
    if (ws_bytes < p.total) return TXE_ERR_WORKSPACE;
    if (G == 0) return TXE_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned* mk = (mask && drop_p > 0.f) ? mask : nullptr;
    const unsigned* dummy_mask = reinterpret_cast<const unsigned*>(X);
    const int mask_ld = (Kt + 31) / 32;
    const float fs = mk ? 1.f / (1.f - drop_p) : 1.f;
    if (n_nodes > 0)
        hipLaunchKernelGGL(gcl_coef_kernel, dim3((n_nodes + 3) / 4), dim3(256), 0, s, rowptr_out, col_dst, n_nodes, norm, pos, pw, coef);
    hipLaunchKernelGGL(cl_wsum_kernel, dim3((G + 3) / 4), dim3(256), 0, s, graph_off, G, pos, pw, wsum, gid);
    const int rc_z = cl_zsum_launch(graph_off, G, n_nodes, X, Kp, mk, dummy_mask, mask_ld, fs, (const float*)coef, (const float*)wsum, Z, s);
    if (rc_z) return rc_z;
    if (!hg) return TXE_OK;          // (the caller folds hg = Z W + b into what consumes it: txe_bilinear_folded_*, wf_by_k)
    VMat A = vmat_plain(Z, Kp, G, Kp);
    VMat B = vmat_plain(Wp, Fop, Kp, Fop);
    Epi E = epi_plain(hg, ld_hg, Fo);
    E.alg_flops = 2.0 * G * (double)Fo * Kt;
    int rc = gemm_nn(A, B, E, G, Fo, Kp, 1, s, p.tail, p.tail_bytes);
    if (rc) return rc;
    if (bias) {
        const long long n = (long long)G * Fo;
        hipLaunchKernelGGL(gcl_add_bias_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, hg, ld_hg, G, Fo,
                           bias);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

// d_hg [G][Fo] -> d_X [N][Kp] (layout of txe_gcn_dense_bwd), dW [Kt][Fo], d_b [Fo] (or NULL), dP, d_pw.
int txe_gcn_collapse_bwd(const int* rowptr_in, const int* col_src, const int* graph_off, int n_nodes, int G, const float* X, int Kh, int Pd,
                         const int* pos, int vocab, const float* Wp, int Fo, float drop_p, const unsigned* mask, const float* norm,
                         const float* pw, const float* coef, const float* wsum, const int* gid, const float* Z, const float* d_hg,
                         long long ld_dhg, int act_on, float act_slope, float* d_X, float* dW, float* d_b, float* dP, float* d_pw, int dz_given,
                         void* ws, size_t ws_bytes, void* stream) {
    // dz_given: `d_hg` IS dZ [G][Kp] (ld_dhg == Kp) -- whoever consumed Z folded hg = Z W + b into its own products (txe_bilinear_folded_*,
    // wf_by_k) and formed dW / d_b itself: no product here, dW / d_b are not written
    if (n_nodes < 0 || G < 0 || Kh < 1 || Pd < 0 || Fo < 1 || !rowptr_in || !graph_off || !X || !Wp || !norm || !coef || !wsum || !gid || !Z ||
        !d_hg || !d_X || (!dW && !dz_given) || !ws)
        return TXE_ERR_ARG;
    if (dz_given && ld_dhg != round_up(Kh + Pd, 32)) return TXE_ERR_ARG;
    if ((Pd > 0 || pw) && (!pos || vocab < 1 || vocab > MAX_VOCAB)) return TXE_ERR_ARG;
    if ((Pd > 0 && !dP) || (pw && !d_pw)) return TXE_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return TXE_ERR_ARG;
    const int Kt = Kh + Pd, Kp = round_up(Kt, 32), Fop = round_up(Fo, 32);
    GclWs p = plan_gcl_ws(ws, n_nodes, G, Kp, Fop, Pd, vocab);
    if (ws_bytes < p.total) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const unsigned* mk = (mask && drop_p > 0.f) ? mask : nullptr;
    const unsigned* dummy_mask = reinterpret_cast<const unsigned*>(X);
    const int mask_ld = (Kt + 31) / 32;
    const float fs = mk ? 1.f / (1.f - drop_p) : 1.f;
    int rc;
    if (dz_given) p.dZ = const_cast<float*>(d_hg);
    if (!dz_given) {   // dZ[g][k] = sum_f d_hg[g][f] Wp[k][f]
        VMat A = vmat_plain(d_hg, ld_dhg, G, Fo);
        VMat B = vmat_plain(Wp, Fop, round_up(Kp, 128), Fop);
        Epi E = epi_plain(p.dZ, Kp, Kp);
        E.alg_flops = 2.0 * G * (double)Kt * Fo;
        rc = gemm_nt(A, B, E, G, Kp, Fo, 1, s, p.tail, p.tail_bytes);
        if (rc) return rc;
    }
    if (!dz_given) {   // dW[k][f] = sum_g Z[g][k] d_hg[g][f]
        VMat A = vmat_plain(Z, Kp, G, Kp);
        VMat B = vmat_plain(d_hg, ld_dhg, G, Fo);
        Epi E = epi_plain(p.part, Fop, Fo);
        E.split_stride = (long long)Kp * Fop;
        E.alg_flops = 2.0 * Kt * (double)Fo * G;
        rc = gemm_tn(A, B, E, Kp, Fo, G, p.splits, s);
        if (rc) return rc;
        const long long n = (long long)Kt * Fo;
        hipLaunchKernelGGL(reduce_splits_sub_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s,
                           (const float*)p.part, G > 0 ? p.splits : 0, E.split_stride, Kt, Fo, Fop, dW);
        TXE_CHECK_LAUNCH();
    }
    if (d_b && !dz_given) {
        rc = colsum_launch(d_hg, ld_dhg, G, Fo, p.cpart, d_b, s);
        if (rc) return rc;
    }
    if (G > 0 && n_nodes > 0) {
        const int nb = (n_nodes + 3) / 4;
        const int ntile = (Kp / 4 + 63) / 64;
        if (pw) {
            hipLaunchKernelGGL(cl_bwd_ds_kernel, dim3((G + 3) / 4), dim3(256), 0, s, G, Kp, (const float*)p.dZ, Z, wsum, p.dS);
            // (the bias makes hg != Z W here, so dS keeps its own kernel: no leading dS workgroups)
            rc = cl_bwd_dot_launch(n_nodes, gid, X, Kp, mk, dummy_mask, mask_ld, fs, (const float*)p.dZ, wsum, coef, p.dc, p.cn, 0, 0, 0, nullptr, 0LL, nullptr,
                                   0LL, nullptr, 4.0 * (n_nodes + (double)G) * Kp, s);
            if (rc) return rc;
            hipLaunchKernelGGL(gcl_bwd_w_kernel, dim3(nb), dim3(256), 0, s, rowptr_in, col_src, n_nodes, norm, pos, pw, (const float*)p.dc,
                               (const float*)p.dS, gid, p.dwv);
        } else {
            hipLaunchKernelGGL(gcl_cn_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, n_nodes, gid, coef, wsum, p.cn);
        }
        {
            const long long nwaves = (long long)((n_nodes + CL_CHUNK - 1) / CL_CHUNK) * ntile;
            ProfScope prof(mk ? "cl_bwd_dx_kernel<true, false>" : "cl_bwd_dx_kernel<false, false>", s, 4.0 * (2.0 * n_nodes + G) * Kp, 1);
            if (mk) hipLaunchKernelGGL((cl_bwd_dx_kernel<true, false>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, s, n_nodes, ntile, gid, X, Kp,
                                       Kh, mk, mask_ld, fs, (const float*)p.dZ, (const float*)p.cn, (const float*)nullptr,
                                       (const float*)nullptr, (const float*)nullptr, act_on, act_slope, d_X, (float*)nullptr);
            else hipLaunchKernelGGL((cl_bwd_dx_kernel<false, false>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, s, n_nodes, ntile, gid, X,
                                    Kp, Kh, dummy_mask, mask_ld, fs, (const float*)p.dZ, (const float*)p.cn, (const float*)nullptr,
                                    (const float*)nullptr, (const float*)nullptr, act_on, act_slope, d_X, (float*)nullptr);
        }
        TXE_CHECK_LAUNCH();
    }
    if (Pd > 0) {
        if (n_nodes > 0)
            hipLaunchKernelGGL(pos_segsum_stage1, dim3(p.seg_blocks), dim3(256), 0, s, (const float*)(d_X + Kh), (long long)Kp, pos, n_nodes, Pd,
                               vocab, p.seg_rows, p.ppart);
        hipLaunchKernelGGL(pos_segsum_stage2, dim3((vocab * Pd + 63) / 64), dim3(256), 0, s, (const float*)p.ppart,
                           n_nodes > 0 ? p.seg_blocks : 0, vocab, Pd, dP);
    }
    if (pw) {
        if (n_nodes > 0)
            hipLaunchKernelGGL(pos_segsum_stage1, dim3(p.seg_blocks), dim3(256), 0, s, (const float*)p.dwv, (long long)1, pos, n_nodes, 1, vocab,
                               p.seg_rows, p.ppart2);
        hipLaunchKernelGGL(pos_segsum_stage2, dim3((vocab + 63) / 64), dim3(256), 0, s, (const float*)p.ppart2, n_nodes > 0 ? p.seg_blocks : 0,
                           vocab, 1, d_pw);
    }
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"

