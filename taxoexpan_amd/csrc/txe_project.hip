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
#include "txe_gemm.h"

namespace txe {

constexpr int MAX_VOCAB = 8;

// wa[h][k]   = sum_d attn_l[h*D+d] * W[(h*D+d)*ldw + k]
// wa[H+h][k] = sum_d attn_r[h*D+d] * W[(h*D+d)*ldw + k]            (k < Kt)
// One workgroup per (row r, 64-column chunk): 64 columns x 16 d-groups, LDS tree over the d-groups.
constexpr int FOLD_DG = 16;
__global__ __launch_bounds__(64 * FOLD_DG) void fold_attn_kernel(const float* __restrict__ W, long long ldw, int Kt,
                                                                 const float* __restrict__ attn_l, const float* __restrict__ attn_r,
                                                                 int H, int D, float* __restrict__ wa) {
    __shared__ float red[FOLD_DG][64];
    const int r = blockIdx.y;                     // 0 .. 2H-1
    const int h = r % H;
    const float* attn = ((r < H) ? attn_l : attn_r) + (long long)h * D;
    const float* Wh = W + (long long)h * D * ldw;
    const int kl = threadIdx.x & 63, dg = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
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
        wa[(long long)r * Kt + k] = s;
    }
}

// dwa[r][k] = sum_s part[s][F + r][k]     r < 2H
__global__ void reduce_ext_rows_kernel(const float* __restrict__ part, int S, long long split_stride, int F, int H2, int Kt,
                                       float* __restrict__ dwa) {
    const int r = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kt) return;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += part[(long long)s * split_stride + (long long)(F + r) * Kt + k];
    dwa[(long long)r * Kt + k] = acc;
}

// One workgroup per weight row f = h*D + d:
//   dW[f][k]    = sum_s part[s][f][k] + attn_l[f] * dwa[h][k] + attn_r[f] * dwa[H+h][k]
//   d_attn_l[f] = sum_k dwa[h][k]   * W[f][k]
//   d_attn_r[f] = sum_k dwa[H+h][k] * W[f][k]
__global__ __launch_bounds__(256) void gat_unfold_kernel(const float* __restrict__ part, int S, long long split_stride,
                                                         const float* __restrict__ dwa, const float* __restrict__ W,
                                                         long long ldw, const float* __restrict__ attn_l,
                                                         const float* __restrict__ attn_r, int H, int D, int Kt,
                                                         float* __restrict__ dW, long long ld_dw,
                                                         float* __restrict__ d_attn_l, float* __restrict__ d_attn_r) {
    __shared__ float red[2][4];
    const int f = blockIdx.x, h = f / D;
    const float al = attn_l[f], ar = attn_r[f];
    float dl = 0.f, dr = 0.f;
    for (int k = threadIdx.x; k < Kt; k += blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += part[(long long)s * split_stride + (long long)f * Kt + k];
        const float gl = dwa[(long long)h * Kt + k], gr = dwa[(long long)(H + h) * Kt + k];
        dW[(long long)f * ld_dw + k] = acc + al * gl + ar * gr;
        const float wv = W[(long long)f * ldw + k];
        dl = fmaf(gl, wv, dl);
        dr = fmaf(gr, wv, dr);
    }
    dl = wave_sum(dl);
    dr = wave_sum(dr);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = dl; red[1][w] = dr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        d_attn_l[f] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        d_attn_r[f] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
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
__global__ __launch_bounds__(256) void pos_segsum_stage1(const float* __restrict__ x, long long ldx, const int* __restrict__ pos,
                                                         int n_rows, int cols, int vocab, int rows_per_block,
                                                         float* __restrict__ part /*[nb][vocab][cols]*/) {
    __shared__ float red[4][MAX_VOCAB][64];
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
    const int jl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    for (int j0 = 0; j0 < cols; j0 += 64) {
        const int j = j0 + jl;
        const int jc = (j < cols) ? j : 0;
        float acc[MAX_VOCAB];
#pragma unroll
        for (int c = 0; c < MAX_VOCAB; ++c) acc[c] = 0.f;
#pragma unroll 4
        for (int m = r0 + rg; m < r1; m += 4) {
            const int pc = pos[m];
            const float v = x[(long long)m * ldx + jc];
#pragma unroll
            for (int c = 0; c < MAX_VOCAB; ++c) acc[c] += (pc == c) ? v : 0.f;
        }
#pragma unroll
        for (int c = 0; c < MAX_VOCAB; ++c) red[rg][c][jl] = acc[c];
        __syncthreads();
        if (rg == 0 && j < cols)
            for (int c = 0; c < vocab; ++c)
                part[((long long)blockIdx.x * vocab + c) * cols + j] = red[0][c][jl] + red[1][c][jl] + red[2][c][jl] + red[3][c][jl];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void pos_segsum_stage2(const float* __restrict__ part, int nb, int vocab, int cols,
                                                         float* __restrict__ out) {
    __shared__ float red[4][64];
    const int il = threadIdx.x & 63, bg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    const int n = vocab * cols;
    const int ic = (i < n) ? i : 0;
    float acc = 0.f;
#pragma unroll 4
    for (int b = bg; b < nb; b += 4) acc += part[(long long)b * n + ic];
    red[bg][il] = acc;
    __syncthreads();
    if (bg == 0 && i < n) out[i] = red[0][il] + red[1][il] + red[2][il] + red[3][il];
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static VMat make_xcat(const float* h, long long ld_h, int n, int Kh, const int* pos, const float* P, int Pd, float drop_p,
                      const unsigned* mask) {
    VMat m = vmat_plain(h, ld_h, n, Kh + Pd);
    m.cols_main = Kh;
    m.p2 = P; m.ld2 = Pd; m.pos = pos;
    vmat_set_mask(m, mask, drop_p);
    return m;
}

__global__ void dropout_mask_kernel(long long n_words, unsigned long long seed, unsigned thr16, unsigned* __restrict__ mask) {
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (long long)gridDim.x * blockDim.x)
        mask[w] = drop_mask_word(seed, (unsigned long long)w, thr16);
}

struct ProjectWs {
    float* wa;      // [2H][Kt]
    float* dwa;     // [2H][Kt]
    float* dxp;     // [N][Pd]
    float* ppart;   // [nb][vocab][Pd]
    float* part;    // [S][(F+2H)][Kt]
    void* tail;     // GEMM tail-splitting workspace
    size_t tail_bytes;
    int splits, seg_blocks, seg_rows;
    size_t total;
};

static ProjectWs plan_ws(void* ws, int n, int F, int H2, int Kt, int Pd, int vocab) {
    ProjectWs p;
    char* b = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) { float* r = (float*)(b + off); off += align_up(bytes, 256); return r; };
    p.wa = take((size_t)H2 * Kt * 4);
    p.dwa = take((size_t)H2 * Kt * 4);
    p.dxp = take((size_t)n * (Pd > 0 ? Pd : 1) * 4);
    p.seg_rows = 64;
    p.seg_blocks = (n + p.seg_rows - 1) / p.seg_rows;
    if (p.seg_blocks < 1) p.seg_blocks = 1;
    p.ppart = take((size_t)p.seg_blocks * (vocab > 0 ? vocab : 1) * (Pd > 0 ? Pd : 1) * 4);
    p.splits = choose_splits(F + H2, Kt, n);
    p.part = take((size_t)p.splits * (F + H2) * Kt * 4);
    p.tail_bytes = gemm_tail_ws_bytes();
    p.tail = take(p.tail_bytes);
    p.total = off;
    return p;
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

// forward workspace: the folded attention rows wa [2H][Kh+Pd] (+ the GEMM tail-splitting scratch, optional: with less than
// this the projection still runs, without tail splitting)
size_t txe_gat_project_fwd_ws_bytes(int Kh, int Pd, int H) {
    return align_up((size_t)2 * H * (Kh + Pd) * 4, 256) + gemm_tail_ws_bytes();
}

size_t txe_gat_project_ws_bytes(int n_nodes, int Kh, int Pd, int H, int D, int vocab) {
    return plan_ws(nullptr, n_nodes, H * D, 2 * H, Kh + Pd, Pd, vocab).total;
}

int txe_gat_project_fwd(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd,
                        const float* W, const float* attn_l, const float* attn_r, int H, int D, float feat_drop_p,
                        const unsigned* mask, float* ft, float* a_ext, void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || H < 1 || D < 1 || !h || !W || !attn_l || !attn_r || !ft || !a_ext || !ws)
        return TXE_ERR_ARG;
    if (Pd > 0 && (!pos || !P)) return TXE_ERR_ARG;
    if (feat_drop_p < 0.f || feat_drop_p >= 1.f) return TXE_ERR_ARG;
    const int F = H * D, H2 = 2 * H, Kt = Kh + Pd;
    if (ws_bytes < (size_t)H2 * Kt * 4) return TXE_ERR_WORKSPACE;
    if (n_nodes == 0) return TXE_OK;
    hipStream_t s = (hipStream_t)stream;
    float* wa = (float*)ws;
    hipLaunchKernelGGL(fold_attn_kernel, dim3((Kt + 63) / 64, H2), dim3(64 * FOLD_DG), 0, s, W, (long long)Kt, Kt, attn_l, attn_r, H, D, wa);
    TXE_CHECK_LAUNCH();
    VMat A = make_xcat(h, ld_h, n_nodes, Kh, pos, P, Pd, feat_drop_p, mask);
    VMat B = vmat_plain(W, Kt, F + H2, Kt);
    B.rows_main = F; B.p3 = wa; B.ld3 = Kt;
    Epi E = epi_plain(ft, F, F);
    E.c2 = a_ext; E.ldc2 = H2;
    const size_t wa_bytes = align_up((size_t)H2 * Kt * 4, 256);
    void* tail = (ws_bytes >= wa_bytes + gemm_tail_ws_bytes()) ? (void*)((char*)ws + wa_bytes) : nullptr;
    return gemm_nt(A, B, E, n_nodes, F + H2, Kt, 1, s, tail, tail ? ws_bytes - wa_bytes : 0);
}

// d_h may be NULL (first layer: the input features carry no gradient).  When d_h is written and act_src is non-NULL
// the result is multiplied by leaky'(act_src[m][k]) -- the backward of the inter-layer activation that produced h.
int txe_gat_project_bwd(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd, int vocab,
                        const float* W, const float* attn_l, const float* attn_r, int H, int D, float feat_drop_p,
                        const unsigned* mask, const float* d_ft, const float* d_a_ext, float* d_h, long long ld_dh,
                        const float* act_src, long long ld_act, float act_slope, float* dW, float* d_attn_l, float* d_attn_r,
                        float* dP, void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || H < 1 || D < 1 || !h || !W || !attn_l || !attn_r || !d_ft || !d_a_ext || !dW ||
        !d_attn_l || !d_attn_r || !ws)
        return TXE_ERR_ARG;
    if (Pd > 0 && (!pos || !P || !dP || vocab < 1 || vocab > MAX_VOCAB)) return TXE_ERR_ARG;
    if (feat_drop_p < 0.f || feat_drop_p >= 1.f) return TXE_ERR_ARG;
    const int F = H * D, H2 = 2 * H, Kt = Kh + Pd;
    ProjectWs p = plan_ws(ws, n_nodes, F, H2, Kt, Pd, vocab);
    if (ws_bytes < p.total) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    hipLaunchKernelGGL(fold_attn_kernel, dim3((Kt + 63) / 64, H2), dim3(64 * FOLD_DG), 0, s, W, (long long)Kt, Kt, attn_l, attn_r, H, D, p.wa);
    TXE_CHECK_LAUNCH();

    VMat G = vmat_plain(d_ft, F, n_nodes, F + H2);          // [d_ft | d_a_ext]
    G.cols_main = F; G.p2 = d_a_ext; G.ld2 = H2;

    // ---- dX = G * Wext, only the columns somebody needs: [c0, Kt) ----
    const int c0 = d_h ? 0 : Kh;
    if (Kt - c0 > 0 && n_nodes > 0) {
        VMat B = vmat_plain(W + c0, Kt, F + H2, Kt - c0);
        B.rows_main = F; B.p3 = p.wa + c0; B.ld3 = Kt;
        Epi E = epi_plain(d_h, ld_dh, Kh - c0);
        E.c2 = p.dxp; E.ldc2 = Pd;
        epi_set_mask(E, mask, Kt, c0, feat_drop_p);
        if (d_h) epi_set_act(E, act_src, ld_act, act_slope);
        rc = gemm_nn(G, B, E, n_nodes, Kt - c0, F + H2, 1, s, p.tail, p.tail_bytes);
        if (rc) return rc;
    }
    // ---- dP[c][j] = sum_{pos[m]==c} dXcat[m][Kh+j] ----
    if (Pd > 0) {
        if (n_nodes > 0) {
            hipLaunchKernelGGL(pos_segsum_stage1, dim3(p.seg_blocks), dim3(256), 0, s, (const float*)p.dxp, (long long)Pd, pos,
                               n_nodes, Pd, vocab, p.seg_rows, p.ppart);
            TXE_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(pos_segsum_stage2, dim3((vocab * Pd + 63) / 64), dim3(256), 0, s, (const float*)p.ppart,
                           n_nodes > 0 ? p.seg_blocks : 0, vocab, Pd, dP);
        TXE_CHECK_LAUNCH();
    }
    // ---- dWext = G^T * Xcat  (split-K over the node dimension) ----
    {
        VMat X = make_xcat(h, ld_h, n_nodes, Kh, pos, P, Pd, feat_drop_p, mask);
        Epi E = epi_plain(p.part, Kt, Kt);
        E.split_stride = (long long)(F + H2) * Kt;
        rc = gemm_tn(G, X, E, F + H2, Kt, n_nodes, p.splits, s);
        if (rc) return rc;
        const int S = n_nodes > 0 ? p.splits : 0;
        hipLaunchKernelGGL(reduce_ext_rows_kernel, dim3((Kt + 127) / 128, H2), dim3(128), 0, s, (const float*)p.part, S,
                           E.split_stride, F, H2, Kt, p.dwa);
        TXE_CHECK_LAUNCH();
        hipLaunchKernelGGL(gat_unfold_kernel, dim3(F), dim3(256), 0, s, (const float*)p.part, S, E.split_stride,
                           (const float*)p.dwa, W, (long long)Kt, attn_l, attn_r, H, D, Kt, dW, (long long)Kt, d_attn_l, d_attn_r);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

// ---------------------------------------------------------------------------------------------
// GCN projection (model_zoo.py:35-37):  hw = dropout(cat(x, P[pos])) @ W,  W is [Kt][Fo] row-major.
// ---------------------------------------------------------------------------------------------
size_t txe_gcn_project_ws_bytes(int n_nodes, int Kh, int Pd, int Fo, int vocab) {
    return plan_ws(nullptr, n_nodes, Kh + Pd, 0, Fo, Pd, vocab).total;
}

int txe_gcn_project_fwd(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd,
                        const float* W, int Fo, float drop_p, const unsigned* mask, float* hw, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || Fo < 1 || !h || !W || !hw) return TXE_ERR_ARG;
    if (Pd > 0 && (!pos || !P)) return TXE_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    const int Kt = Kh + Pd;
    VMat A = make_xcat(h, ld_h, n_nodes, Kh, pos, P, Pd, drop_p, mask);
    VMat B = vmat_plain(W, Fo, Kt, Fo);
    Epi E = epi_plain(hw, Fo, Fo);
    return gemm_nn(A, B, E, n_nodes, Fo, Kt, 1, (hipStream_t)stream);
}

int txe_gcn_project_bwd(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd, int vocab,
                        const float* W, int Fo, float drop_p, const unsigned* mask, const float* d_hw, float* d_h,
                        long long ld_dh, const float* act_src, long long ld_act, float act_slope, float* dW, float* dP,
                        void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || Kh < 1 || Pd < 0 || Fo < 1 || !h || !W || !d_hw || !dW || !ws) return TXE_ERR_ARG;
    if (Pd > 0 && (!pos || !P || !dP || vocab < 1 || vocab > MAX_VOCAB)) return TXE_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return TXE_ERR_ARG;
    const int Kt = Kh + Pd;
    ProjectWs p = plan_ws(ws, n_nodes, Kt, 0, Fo, Pd, vocab);   // part: [S][Kt][Fo]
    if (ws_bytes < p.total) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    VMat G = vmat_plain(d_hw, Fo, n_nodes, Fo);
    // dXcat[m][c] = sum_fo d_hw[m][fo] * W[c][fo]   (NT), columns [c0, Kt)
    const int c0 = d_h ? 0 : Kh;
    if (Kt - c0 > 0 && n_nodes > 0) {
        VMat B = vmat_plain(W + (long long)c0 * Fo, Fo, Kt - c0, Fo);
        Epi E = epi_plain(d_h, ld_dh, Kh - c0);
        E.c2 = p.dxp; E.ldc2 = Pd;
        epi_set_mask(E, mask, Kt, c0, drop_p);
        if (d_h) epi_set_act(E, act_src, ld_act, act_slope);
        rc = gemm_nt(G, B, E, n_nodes, Kt - c0, Fo, 1, s, p.tail, p.tail_bytes);
        if (rc) return rc;
    }
    if (Pd > 0) {
        if (n_nodes > 0) {
            hipLaunchKernelGGL(pos_segsum_stage1, dim3(p.seg_blocks), dim3(256), 0, s, (const float*)p.dxp, (long long)Pd, pos,
                               n_nodes, Pd, vocab, p.seg_rows, p.ppart);
            TXE_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(pos_segsum_stage2, dim3((vocab * Pd + 63) / 64), dim3(256), 0, s, (const float*)p.ppart,
                           n_nodes > 0 ? p.seg_blocks : 0, vocab, Pd, dP);
        TXE_CHECK_LAUNCH();
    }
    // dW[kt][fo] = sum_m Xcat[m][kt] * d_hw[m][fo]   (TN, split-K over nodes)
    {
        VMat X = make_xcat(h, ld_h, n_nodes, Kh, pos, P, Pd, drop_p, mask);
        Epi E = epi_plain(p.part, Fo, Fo);
        E.split_stride = (long long)Kt * Fo;
        rc = gemm_tn(X, G, E, Kt, Fo, n_nodes, p.splits, s);
        if (rc) return rc;
        const long long n = (long long)Kt * Fo;
        const int nb = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(reduce_splits_kernel, dim3(nb), dim3(256), 0, s, (const float*)p.part, n_nodes > 0 ? p.splits : 0,
                           E.split_stride, n, dW);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

}  // extern "C"
