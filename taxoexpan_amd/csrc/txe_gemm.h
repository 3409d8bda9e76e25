// fp32 MFMA GEMM for gfx950 with "virtual matrix" operand loaders and fused epilogues.
//
// Every dense product on TaxoExpan's hot path (GAT/GCN feature projections model_zoo.py:37,83,
// their two backward products, the bilinear match model_zoo.py:313,328 and the all-candidate
// scoring loop test_fast.py:116-123) goes through this one kernel:
//
//     C[m][n] = sum_kk  A(m,kk) * B(kk,n)          m < M, n < N, kk in [k0,k1)
//
// * math: v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak on MI355X -- there is no TF32/xf32 on
//   gfx950).  Block tile 128 x BN x 32 (BN = 128: 2x2 waves of 64x64; BN = 64: 4x1 waves of 32x64 -- more,
//   smaller workgroups for narrow outputs and to soften wave quantisation on 256 CUs).  Operands are staged
//   global -> registers -> LDS (double buffered, one barrier per k-tile).
// * operands are *virtual* row-major matrices (VMat): the concat of node features with the position
//   embedding row (model_zoo.py:215), the feature dropout (model_zoo.py:82, as a precomputed bit mask), row /
//   column extensions that carry the folded attention projections ... are synthesised by the loader, so none
//   of those tensors is ever materialised in HBM.
// * either operand may be read "k-contiguous" (A[m][kk], kk fastest) or "row-contiguous"
//   (A(m,kk) = Mat[kk][m]); that covers NT / NN / TN products without transposing in HBM.
// * NO branch between a load and its first use: hipcc waits vmcnt(0) at control-flow merges behind pending loads
//   and schedules only inside basic blocks, which serialised the staging stream at one HBM round trip per vector
//   (measured 25 TF/s).  Hence: two-phase staging (issue = loads only / finish = selects + dropout, after the MFMA
//   block), clamped always-valid addresses instead of guards, dummy-but-readable pointers for absent extras, and a
//   k-loop split by the kind of tile being fetched (plain prefix, generic tail) instead of a per-tile branch.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "txe_common.h"

namespace txe {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Logical row-major matrix [rows][cols] assembled from up to three arrays:
//   cols [0, cols_main)      : p  (rows < rows_main)  or p3 (rows >= rows_main, row r-rows_main)
//   cols [cols_main, cols)   : p2[er*ld2 + c-cols_main], er = pos ? pos[r] : r      (table / 2nd matrix)
// then optionally * (mask bit(r,c) ? drop_scale : 0)  (mask: 32 columns per word, mask_ld words/row).
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
    const unsigned* mask;      // ALWAYS a readable address (the operand itself when there is no dropout): the loader
    long long mask_ld;         // fetches one word per vector unconditionally, `mask_on` decides whether it is applied
    int mask_on;
    float drop_scale;
};

static inline VMat vmat_plain(const float* p, long long ld, int rows, int cols) {
    VMat m;
    m.p = p; m.ld = ld; m.rows = rows; m.cols = cols; m.cols_main = cols;
    m.p2 = nullptr; m.ld2 = 0; m.pos = nullptr;
    m.rows_main = rows; m.p3 = nullptr; m.ld3 = 0;
    m.mask = reinterpret_cast<const unsigned*>(p); m.mask_ld = 0; m.mask_on = 0; m.drop_scale = 1.f;
    return m;
}
static inline void vmat_set_mask(VMat& m, const unsigned* mask, float drop_p) {
    if (mask && drop_p > 0.f) { m.mask = mask; m.mask_ld = (m.cols + 31) / 32; m.mask_on = 1; m.drop_scale = 1.f / (1.f - drop_p); }
}

// Output side.  Logical C [rows][cols]:
//   n <  cols_main : c [m*ldc  + n]
//   n >= cols_main : c2[m*ldc2 + n - cols_main]
// value = acc (* (mask bit(m, n+mask_col0) ? drop_scale : 0)) (* leaky'(act_src[m][n]) for n<cols_main) (exp() if apply_exp).
// Split-K: block z writes at c + z*split_stride (no extras expected).
struct Epi {
    float* c;
    long long ldc;
    int cols_main;
    float* c2;
    long long ldc2;
    const float* act_src;      // act_src / mask: always readable addresses; act_on / mask_on say whether they apply
    long long ld_act;
    float act_slope;
    int act_on;
    const unsigned* mask;
    long long mask_ld;
    int mask_col0;
    int mask_on;
    float drop_scale;
    int apply_exp;
    long long split_stride;
    // count mode (fused ranking of the scoring loop): nothing is stored; for row m and each p in [cnt_off[m], cnt_off[m+1]):
    //   cnt_out[p] += #{ n < N : value(m, n) > cnt_thr[p] }  (cnt_mode 1)  /  < cnt_thr[p]  (cnt_mode 2)    -- int32 atomics, exact
    // pick mode (cnt_mode 3; the thresholds of that ranking): column n belongs to the row m with cnt_off[m] <= n < cnt_off[m+1] (the
    //   columns are the gathered true parents of the rows' queries, query by query); only c[n] = value(m, n) of those pairs is stored
    //   and tiles that hold none of them return at once -- a staircase of ~(rows/128 + columns/BN) tiles instead of the whole product
    // top-k mode (cnt_mode 4: larger is better, 5: smaller is better; the scoring loop's "best k parents", infer.py:100-106): nothing of
    //   the product is stored; every tile leaves, per row, its topk_k best columns in the order Python's stable sorted() gives them
    //   (better value first, equal values by ascending column; NaN ranks last) as keys (value, negated in mode 5) + column numbers at
    //   topk_key / topk_idx [M][column tiles][topk_k]; a merge kernel (txe_topk_merge) picks each row's best k among the tiles' lists
    int cnt_mode;
    int topk_k;
    float* topk_key;
    int* topk_idx;
    int* topk_floor;           // [M] ordered-int keys (topk_ord): a lower bound of each row's final k-th best key, raised by every tile
    int force_bn128;           // 1: 128-wide column tiles whatever the shape (the top-k scratch is laid out by them)
    const int* cnt_off;
    const float* cnt_thr;
    int* cnt_out;
    // profiling only: the product's ALGORITHMIC flops when the operands carry tile padding (0: 2*M*N*K as launched)
    double alg_flops;
    // 1: every element is accumulated in plain k order whatever the tile count (no k-split of a last partial round): the scoring
    // loop compares scores of different launches bit for bit
    int plain_k_order;
    // test / micro-benchmark routes (txe_gemm_plain's `route` argument; 0 everywhere in the model paths): GEMM_ROUTE_* bits
    int route;
    // 0, or the caller's promise that columns [k_valid, K) of BOTH k-contiguous operands are zero padding: the persistent kernel
    // then skips the MFMA steps of a tile's last k-tile that would only add 0 * 0 (K = 300 padded to 320: 2 of its 4 eight-column
    // groups, 5 % of the product's matrix work; bit-identical sums)
    int k_valid;
};
constexpr int GEMM_ROUTE_NO_PERSIST = 1;     // whole rounds stay on gemm_kernel (the persistent kernel's tiles are bit-identical)
constexpr int GEMM_ROUTE_NO_TN_LDS = 2;      // split-K TN products on gemm_kernel<false,false,4,4,160> (same k order per element)
constexpr int GEMM_ROUTE_FORCE_BN160 = 4;    // every eligible split-K product on 128 x 160 tiles, whatever its size

static inline Epi epi_plain(float* c, long long ldc, int cols) {
    Epi e;
    e.c = c; e.ldc = ldc; e.cols_main = cols; e.c2 = nullptr; e.ldc2 = 0;
    e.act_src = c; e.ld_act = 0; e.act_slope = 1.f; e.act_on = 0;
    e.mask = reinterpret_cast<const unsigned*>(c); e.mask_ld = 0; e.mask_col0 = 0; e.mask_on = 0; e.drop_scale = 1.f;
    e.apply_exp = 0; e.split_stride = 0;
    e.cnt_mode = 0; e.cnt_off = nullptr; e.cnt_thr = nullptr; e.cnt_out = nullptr;
    e.topk_k = 0; e.topk_key = nullptr; e.topk_idx = nullptr; e.topk_floor = nullptr; e.force_bn128 = 0;
    e.alg_flops = 0.0; e.route = 0; e.k_valid = 0;
    e.plain_k_order = 0;
    return e;
}
static inline void epi_set_mask(Epi& e, const unsigned* mask, int total_cols, int col0, float drop_p) {
    if (mask && drop_p > 0.f) { e.mask = mask; e.mask_ld = (total_cols + 31) / 32; e.mask_col0 = col0; e.mask_on = 1; e.drop_scale = 1.f / (1.f - drop_p); }
}
static inline void epi_set_act(Epi& e, const float* act_src, long long ld_act, float slope) {
    if (act_src) { e.act_src = act_src; e.ld_act = ld_act; e.act_slope = slope; e.act_on = 1; }
}

// ---- staging, phase 1: loads only ----------------------------------------------------------------------------
// FAST: the whole tile lies inside the plain column range (c + V <= cols_main); only rows can be out of range.
// Otherwise the element path: per element ONE load from a selected (always valid) address.
template <int V, bool FAST>
__device__ __forceinline__ void vmat_issue(const VMat& M, int r, int c, long long er, float* v, unsigned& mw) {
    const int rr = (r < M.rows) ? r : 0;
    const float* row = (rr < M.rows_main) ? (M.p + (long long)rr * M.ld) : (M.p3 + (long long)(rr - M.rows_main) * M.ld3);
    if constexpr (FAST) {
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
        const float* ext = M.p2 ? (M.p2 + er * M.ld2) : row;          // block-uniform select
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int cc = c + e;
            const bool in_main = cc < M.cols_main;
            const bool in_ext = (!in_main) & (cc < M.cols) & (M.p2 != nullptr);
            const float* a = in_main ? (row + cc) : (in_ext ? (ext + (cc - M.cols_main)) : row);
            v[e] = *a;
        }
    }
    mw = M.mask[(long long)rr * M.mask_ld + (M.mask_on ? ((c < M.cols ? c : 0) >> 5) : 0)];
}

// ---- staging, phase 2: bounds selects + dropout factor (runs after the MFMA block) ---------------------------
template <int V, bool FAST>
__device__ __forceinline__ void vmat_finish(const VMat& M, int r, int c, float* v, unsigned mw) {
    const bool rok = r < M.rows;
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int cc = c + e;
        bool ok = rok;
        if constexpr (!FAST) ok = rok & ((cc < M.cols_main) | ((cc < M.cols) & (M.p2 != nullptr)));
        const unsigned keep = ((mw >> (cc & 31)) & 1u) | (M.mask_on ? 0u : 1u);      // bitwise: no short-circuit branches
        v[e] = (ok & (keep != 0u)) ? v[e] * M.drop_scale : 0.f;
    }
}

constexpr int GEMM_BM = 128, GEMM_BK = 32, GEMM_THREADS = 256;
constexpr int GEMM_KPAD = GEMM_BK + 4;  // k-contiguous LDS row stride (floats): conflict-free ds_read_b128

// An operand tile has ROWS (128 or 64) rows of the GEMM's m/n dimension and BK k-values.
// KC = true : tile is [ROWS][BK] read along k        (LDS [row][BK+4])
// KC = false: tile is [BK][ROWS] read along the rows (LDS [k][ROWS])
template <bool KC, int V, int ROWS> struct StageGeom {
    static constexpr int VPR = (KC ? GEMM_BK : ROWS) / V;        // vectors per tile line
    static constexpr int LINES = KC ? ROWS : GEMM_BK;
    // FLAT: a line's vector count does not divide the workgroup (160-wide row-contiguous tiles: 40 vectors): the thread's vector of
    // pass p is number tid + 256 p of the tile in line-major order; otherwise whole lines per pass (uniform stride between passes)
    static constexpr bool FLAT = (GEMM_THREADS % VPR) != 0;
    static constexpr int LPP = GEMM_THREADS / VPR;               // lines per pass (not FLAT)
    static constexpr int PASSES = LINES * VPR / GEMM_THREADS;
    static_assert((LINES * VPR) % GEMM_THREADS == 0, "tile vectors per thread");
    static constexpr int NREG = PASSES * V;
    static constexpr int LDS = KC ? ROWS * GEMM_KPAD : GEMM_BK * ROWS;
};

template <bool KC, int V, int ROWS>
__device__ __forceinline__ void stage_lq(int p, int& line, int& q) {
    using G = StageGeom<KC, V, ROWS>;
    const int t = threadIdx.x;
    if constexpr (G::FLAT) { const int v = t + p * GEMM_THREADS; q = v % G::VPR; line = v / G::VPR; }
    else { q = t % G::VPR; line = t / G::VPR + p * G::LPP; }
}

template <bool KC, int V, int ROWS>
__device__ __forceinline__ void stage_coord(int row0, int k0, int p, int& r, int& c) {
    int line, q;
    stage_lq<KC, V, ROWS>(p, line, q);
    if constexpr (KC) { r = row0 + line; c = k0 + q * V; }
    else { r = k0 + line; c = row0 + q * V; }
}

template <bool KC, int V, int ROWS, bool FAST>
__device__ __forceinline__ void stage_issue(const VMat& M, int row0, int k0, float* regs, unsigned* mws) {
    using G = StageGeom<KC, V, ROWS>;
    long long er[G::PASSES];
#pragma unroll
    for (int p = 0; p < G::PASSES; ++p) {
        int r, c;
        stage_coord<KC, V, ROWS>(row0, k0, p, r, c);
        const int rr = (r < M.rows) ? r : 0;
        er[p] = rr;
        if constexpr (!FAST) { if (M.pos) er[p] = M.pos[rr]; }      // all position loads first, one wait
    }
#pragma unroll
    for (int p = 0; p < G::PASSES; ++p) {
        int r, c;
        stage_coord<KC, V, ROWS>(row0, k0, p, r, c);
        vmat_issue<V, FAST>(M, r, c, er[p], regs + p * V, mws[p]);
    }
}

template <bool KC, int V, int ROWS, bool FAST>
__device__ __forceinline__ void stage_finish(const VMat& M, int row0, int k0, float* regs, const unsigned* mws) {
    using G = StageGeom<KC, V, ROWS>;
#pragma unroll
    for (int p = 0; p < G::PASSES; ++p) {
        int r, c;
        stage_coord<KC, V, ROWS>(row0, k0, p, r, c);
        vmat_finish<V, FAST>(M, r, c, regs + p * V, mws[p]);
    }
}

// ---- plain-tile fast path with HOISTED addressing -------------------------------------------------------------------
// PMC on the 4096^3 product showed 278 VALU instructions per k-tile per wave -- mostly 64-bit address arithmetic, clamps
// and selects -- sitting between the MFMA blocks of both co-resident waves at the same time (~27 % idle matrix pipe).
// For a workgroup whose 128 (64) rows are all valid and from one source array, and for k-tiles inside the plain range, the
// address of pass p of a thread is  base + p * pass_stride,  and base moves by a uniform stride per tile: one 64-bit
// pointer per operand lives in registers, no clamps, no selects, and nothing to do in the finish phase unless the
// operand carries a dropout mask.
template <bool KC, int ROWS>
__device__ __forceinline__ bool block_is_plain(const VMat& M, int row0) {
    if constexpr (KC) {       // rows = the GEMM's m/n range of this workgroup: all from p or all from p3.  A ragged LAST panel is plain
                              // too: the passes past the operand re-read a valid row (fast_issue<..., CLAMP>), and the product rows /
                              // columns they feed are never stored.  (Left to the generic loader, the one ragged row panel of a
                              // 24,736-row product made its two workgroups the stragglers of the launch: 110 us against 86 us for
                              // the LARGER 32,768-row product.)
        const int rend = (row0 + ROWS < M.rows) ? row0 + ROWS : M.rows;
        return (row0 < M.rows) && ((rend <= M.rows_main) || (row0 >= M.rows_main));
    } else {                  // columns = the m/n range: inside the plain column range; for an operand without an extension a
                              // ragged last tile is fine too (out-of-range column vectors are clamped and zeroed)
        return (M.cols_main == M.cols) ? (row0 < M.cols) : (row0 + ROWS <= M.cols_main);
    }
}

template <bool KC, int V, int ROWS> struct FastPtr {
    const float* base;         // address of pass 0 for the current tile
    const unsigned* mbase;     // mask word of pass 0 for the current tile (only meaningful when mask_on)
    long long pstride, adv;    // elements between passes / per k-tile (block-uniform)
    long long mpstride, madv;  // mask words between passes / per k-tile
    bool cvalid;               // row-contiguous operands: this thread's column vector lies inside the matrix
    // FLAT geometry (row-contiguous operands only): per-pass element / mask-word offsets from pass 0, per-pass column validity
    int off[StageGeom<KC, V, ROWS>::FLAT ? StageGeom<KC, V, ROWS>::PASSES : 1];
    int moff[StageGeom<KC, V, ROWS>::FLAT ? StageGeom<KC, V, ROWS>::PASSES : 1];
    unsigned cvmask;
};

template <bool KC, int V, int ROWS>
__device__ __forceinline__ void fast_init(const VMat& M, int row0, int kbeg, FastPtr<KC, V, ROWS>& f) {
    using G = StageGeom<KC, V, ROWS>;
    int r, c;
    stage_coord<KC, V, ROWS>(row0, kbeg, 0, r, c);
    f.cvalid = KC ? true : (c + V <= M.cols);
    if (!f.cvalid) c = 0;                                // clamped (always readable) column, zeroed in fast_finish
    const int rr = (r < M.rows) ? r : 0;                 // only dereferenced when the block/tile is plain (then r is valid)
    const float* row = (rr < M.rows_main) ? (M.p + (long long)rr * M.ld) : (M.p3 + (long long)(rr - M.rows_main) * M.ld3);
    const long long ld = (rr < M.rows_main) ? M.ld : M.ld3;
    f.base = row + c;
    f.pstride = (long long)G::LPP * ld;
    f.adv = KC ? (long long)GEMM_BK : (long long)GEMM_BK * ld;
    f.mbase = M.mask + (long long)rr * M.mask_ld + (c >> 5);
    f.mpstride = (long long)G::LPP * M.mask_ld;
    f.madv = KC ? 1 : (long long)GEMM_BK * M.mask_ld;
    f.cvmask = 0u;
    if constexpr (KC) {                                  // passes whose row lies inside the operand (fast_issue<..., CLAMP = true>)
        int l0, q0;
        stage_lq<KC, V, ROWS>(0, l0, q0);
#pragma unroll
        for (int p = 0; p < G::PASSES; ++p) f.cvmask |= (row0 + l0 + p * G::LPP < M.rows) ? (1u << p) : 0u;
    }
    if constexpr (G::FLAT) {
        static_assert(!KC, "flat staging serves row-contiguous operands");
        int l0, q0;
        stage_lq<KC, V, ROWS>(0, l0, q0);
#pragma unroll
        for (int p = 0; p < G::PASSES; ++p) {
            int lp, qp;
            stage_lq<KC, V, ROWS>(p, lp, qp);
            const int cp = row0 + qp * V;
            const bool ok = cp + V <= M.cols;
            f.cvmask |= ok ? (1u << p) : 0u;
            const int cc = ok ? cp : 0, c0 = f.cvalid ? row0 + q0 * V : 0;
            f.off[p] = (int)((lp - l0) * ld) + (cc - c0);
            f.moff[p] = (int)((lp - l0) * M.mask_ld) + ((cc >> 5) - (c0 >> 5));
        }
    }
}

// CLAMP (k-contiguous operands of a ragged last row panel): a pass whose row lies past the operand re-reads pass 0's row -- always
// readable, and the product rows it feeds are never stored
template <bool KC, int V, int ROWS, bool CLAMP = false>
__device__ __forceinline__ void fast_issue(const VMat& M, FastPtr<KC, V, ROWS>& f, float* regs, unsigned* mws) {
    using G = StageGeom<KC, V, ROWS>;
#pragma unroll
    for (int p = 0; p < G::PASSES; ++p) {
        const float* a;
        if constexpr (G::FLAT) a = f.base + f.off[p];
        else if constexpr (CLAMP && KC) a = f.base + (((f.cvmask >> p) & 1u) ? p : 0) * f.pstride;
        else a = f.base + p * f.pstride;
        float* v = regs + p * V;
        if constexpr (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(a);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else if constexpr (V == 2) {
            const float2 t = *reinterpret_cast<const float2*>(a);
            v[0] = t.x; v[1] = t.y;
        } else {
            v[0] = *a;
        }
    }
    f.base += f.adv;
    if (M.mask_on) {                                    // block-uniform; only dropout operands pay the extra word loads
#pragma unroll
        for (int p = 0; p < G::PASSES; ++p) {
            if constexpr (G::FLAT) mws[p] = f.mbase[f.moff[p]];
            else if constexpr (CLAMP && KC) mws[p] = f.mbase[(((f.cvmask >> p) & 1u) ? p : 0) * f.mpstride];
            else mws[p] = f.mbase[p * f.mpstride];
        }
        f.mbase += f.madv;
    }
}

template <bool KC, int V, int ROWS>
__device__ __forceinline__ void fast_finish(const VMat& M, const FastPtr<KC, V, ROWS>& f, float* regs, const unsigned* mws) {
    using G = StageGeom<KC, V, ROWS>;
    // (a row-contiguous operand's column vectors past the matrix were read from a clamped, readable address and are NOT zeroed: column
    //  j of the operand only feeds row / column j of the product, which lies past M / N and is never stored -- 24 selects per k-tile
    //  and thread less in the weight-gradient products)
    if (M.mask_on) {
#pragma unroll
        for (int p = 0; p < G::PASSES; ++p) {
            int lp, qp;
            stage_lq<KC, V, ROWS>(p, lp, qp);
            const int bit0 = (qp * V) & 31;                      // column of element 0 inside its mask word (tile starts are x32)
#pragma unroll
            for (int e = 0; e < V; ++e)
                regs[p * V + e] = ((mws[p] >> (bit0 + e)) & 1u) ? regs[p * V + e] * M.drop_scale : 0.f;
        }
    }
}

template <bool KC, int V, int ROWS>
__device__ __forceinline__ void stage_store(float* lds, const float* regs) {
    using G = StageGeom<KC, V, ROWS>;
#pragma unroll
    for (int p = 0; p < G::PASSES; ++p) {
        int line, q;
        stage_lq<KC, V, ROWS>(p, line, q);
        float* d = KC ? (lds + line * GEMM_KPAD + q * V) : (lds + line * ROWS + q * V);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = regs[p * V + e];
    }
}

// MFMA operand fragment for the 32-row sub-tile starting at row r0, k-group kb (8 k values):
// lane l supplies row (l&31) and k = kb*8 + (l>>5)*4 + s for step s = 0..3.  A and B use the same
// (lane-half, step) -> k map, so the products line up whatever the storage mode.
template <bool KC, int ROWS>
__device__ __forceinline__ void frag_load(const float* lds, int r0, int kb, float* f) {
    const int l = threadIdx.x & 63;
    const int row = r0 + (l & 31), kk = kb * 8 + (l >> 5) * 4;
    if constexpr (KC) {
        const float4 t = *reinterpret_cast<const float4*>(lds + row * GEMM_KPAD + kk);
        f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = lds[(kk + s) * ROWS + row];
    }
}

// ---- tail splitting ----------------------------------------------------------------------------------------------------
// With T output tiles and `slots` co-resident workgroups (2 per CU), T = q*slots + r leaves a last round of only r
// workgroups (MAG layer-1 forward: 564 tiles on 512 slots -> the second round runs at 10 % occupancy and doubles the
// kernel time).  The first q*slots workgroups compute whole tiles; each of the r leftover tiles is cut into S = slots/r
// k-slices that all run concurrently in the last round and park raw partial tiles in a workspace; a tiny fix-up kernel
// adds the S slices in fixed order (deterministic) and applies the real epilogue.
struct Tail {
    int nfull;       // workgroups [0, nfull) own whole tiles; S == 0 -> no tail splitting
    int S;           // k-slices per leftover tile
    int ksplit;      // k-range of one slice (multiple of BK)
    float* ws;       // [r*S][GEMM_BM*BN] raw partial tiles
    int row_fast;    // 1: consecutive workgroups walk DOWN a column of tiles (few row panels, many column tiles: the scoring GEMM)
    int tile0;       // this launch owns the tiles [tile0, all) of the product (the leading whole rounds went to gemm_persist_kernel)
};

__device__ __forceinline__ void epi_store_one(const Epi& E, int m, int n, float acc, float* cbase) {
    const bool main_col = n < E.cols_main;
    const int cm = n + E.mask_col0;
    const unsigned wd = E.mask[E.mask_on ? ((long long)m * E.mask_ld + (cm >> 5)) : 0];
    const float av = E.act_src[((E.act_on != 0) & main_col) ? ((long long)m * E.ld_act + n) : 0];
    const unsigned keep = ((wd >> (cm & 31)) & 1u) | (E.mask_on ? 0u : 1u);
    float g = keep ? E.drop_scale : 0.f;
    g *= ((E.act_on != 0) & main_col & !(av > 0.f)) ? E.act_slope : 1.f;
    const float x = acc * g;
    const float val = E.apply_exp ? __expf(x) : x;
    if (main_col) cbase[(long long)m * E.ldc + n] = val;
    else E.c2[(long long)m * E.ldc2 + (n - E.cols_main)] = val;
}

template <int BN>
__global__ __launch_bounds__(256) void gemm_tail_fixup_kernel(const Epi E, const Tail T, const int M, const int N) {
    const int nbn = (N + BN - 1) / BN;
    const int tile = T.tile0 + T.nfull + blockIdx.x;
    const int nbm = (M + GEMM_BM - 1) / GEMM_BM;
    const int m0 = (T.row_fast ? tile % nbm : tile / nbn) * GEMM_BM, n0 = (T.row_fast ? tile / nbm : tile % nbn) * BN;
    const float* part = T.ws + (long long)blockIdx.x * T.S * (GEMM_BM * BN);
    constexpr int CH = GEMM_BM * BN / 16;            // 16 workgroups per leftover tile (blockIdx.y)
    for (int idx = blockIdx.y * CH + threadIdx.x; idx < (blockIdx.y + 1) * CH; idx += 256) {
        const int m = m0 + idx / BN, n = n0 + idx % BN;
        float acc = 0.f;
        for (int z0 = 0; z0 < T.S; z0 += 4) {        // slice order kept; four clamped loads in flight
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = part[(long long)min(z0 + q, T.S - 1) * (GEMM_BM * BN) + idx];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += (z0 + q < T.S) ? v[q] : 0.f;
        }
        if (m < M && n < N) epi_store_one(E, m, n, acc, E.c);
    }
}

// The epilogue of a 128 x BN tile staged in LDS (Cs = smem, row pitch BN + 4; SMEM floats in all: what lies behind the tile is scratch
// for the count mode) -- plain / exp / mask / activation stores, the pick, count and best-k modes.  Shared by gemm_kernel and the
// bf16-pipe product of txe_gemm_split.hip (the scoring loop's four entry points must produce bit-identical scores: ONE epilogue).
// cnt_pb / cnt_np / cnt_th: the count mode's per-row positive range and first thresholds, fetched by the caller before its k-loop.
template <int BN, int SMEM>
__device__ __forceinline__ void gemm_tile_epilogue(const Epi& E, float* smem, const int m0, const int n0, const int M, const int N, const int tn,
                                                   const int nbn, const int zslice, const int cnt_pb, const int cnt_np, const float (&cnt_th)[8]) {
    constexpr int CLD = BN + 4;
    constexpr int C4 = BN / 4;                       // 16-byte chunks per tile row
    float* Cs = smem;
    if (E.cnt_mode == 3) {                           // pick mode: store the rows' own columns only
        for (int idx = threadIdx.x; idx < GEMM_BM * C4; idx += GEMM_THREADS) {
            const int row = idx / C4, c4 = idx % C4;
            const int m = m0 + row, mc = min(m, M - 1);
            const int lo = E.cnt_off[mc], hi = E.cnt_off[mc + 1];
            const float4 t4 = *reinterpret_cast<const float4*>(Cs + row * CLD + c4 * 4);
            const float v4[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + c4 * 4 + q;
                if (m < M && n >= lo && n < hi && n < N) E.c[n] = E.apply_exp ? __expf(v4[q]) : v4[q];
            }
        }
        return;
    }
    if (E.cnt_mode == 4 || E.cnt_mode == 5) {
        // best-k columns of every row of this tile: two threads per row, each scans its half row (ascending columns) into a best-k list,
        // the pair merges, the even thread writes the row's topk_k entries of this column tile
        static_assert(GEMM_THREADS == 2 * GEMM_BM, "two threads per tile row");
        constexpr int HW = BN / 2;
        const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
        const int m = m0 + row, nb = n0 + half * HW;
        const bool larger = E.cnt_mode == 4;
        float bk[TOPK_MAX];
        int bi[TOPK_MAX];
        topk_init(bk, bi);
        // the row's floor: the largest k-th best key any tile of this row has published so far -- a lower bound of the row's FINAL k-th
        // best key, so a value strictly below it can never be selected and is skipped.  After the first round of tiles almost nothing
        // passes, and the (divergent) insert branch is rarely taken by any lane of a wave.  A stale read (another XCD's L2) is only a
        // lower floor: the selection is exact and deterministic whatever the timing; only the work saved varies.
        const int mf = (m < M) ? m : 0;
        const float floor_key = topk_unord(__hip_atomic_load(E.topk_floor + mf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int kk = E.topk_k;
        float wk = -INFINITY;                            // the list's current k-th best: what a candidate has to beat
        int wi = 0x7fffffff;
#pragma unroll 4
        for (int j = 0; j < HW / 4; ++j) {
            const float4 t4 = *reinterpret_cast<const float4*>(Cs + row * CLD + half * HW + 4 * j);
            const float v4[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb + 4 * j + q;
                const float key = topk_key_of(E.apply_exp ? __expf(v4[q]) : v4[q], larger);
                if (n < N && !(key < floor_key) && topk_better(key, n, wk, wi)) {
                    topk_insert(bk, bi, key, n);
                    topk_kth(bk, bi, kk, wk, wi);
                }
            }
        }
        // the odd thread's list goes to the even one (all lanes shuffle; only the even thread's merge is kept)
#pragma unroll
        for (int t = 0; t < TOPK_MAX; ++t) {
            const float ok = __shfl_xor(bk[t], 1, 64);
            const int oi = __shfl_xor(bi[t], 1, 64);
            // (the partner's list is sorted: once an entry fails, the rest would too -- the insert is predicated, not skipped, to keep
            //  the shuffles of the next round uniform)
            if (half == 0 && t < kk && topk_better(ok, oi, wk, wi)) {
                topk_insert(bk, bi, ok, oi);
                topk_kth(bk, bi, kk, wk, wi);
            }
        }
        if (half == 0 && m < M) {
            const long long o = ((long long)m * nbn + tn) * E.topk_k;
            float kth = -INFINITY;
            int kth_i = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < TOPK_MAX; ++t)
                if (t < E.topk_k) { E.topk_key[o + t] = bk[t]; E.topk_idx[o + t] = bi[t]; kth = bk[t]; kth_i = bi[t]; }
            // this tile holds k real entries at or above kth: the row's final k-th best key cannot be lower
            if (kth_i != 0x7fffffff && kth > floor_key) atomicMax(E.topk_floor + m, topk_ord(kth));
        }
        return;
    }
    if (E.cnt_mode != 0) {
        // fused ranking: every thread owns 4 consecutive columns of a row; the C4 lanes of a row reduce their counts by butterfly.
        // The rows' positive ranges and (up to PS) thresholds are staged once per tile in the LDS left over behind the C tile.
        constexpr int FREE = SMEM - GEMM_BM * CLD;
        constexpr int PS = (FREE - 2 * GEMM_BM) / GEMM_BM >= 8 ? 8 : ((FREE - 2 * GEMM_BM) / GEMM_BM > 0 ? (FREE - 2 * GEMM_BM) / GEMM_BM : 0);
        float* s_thr = smem + GEMM_BM * CLD;
        int* s_pb = reinterpret_cast<int*>(s_thr + GEMM_BM * PS);
        int* s_np = s_pb + GEMM_BM;
        if constexpr (PS > 0) {
            if (threadIdx.x < GEMM_BM) {
                s_pb[threadIdx.x] = cnt_pb;
                s_np[threadIdx.x] = cnt_np;
#pragma unroll
                for (int k = 0; k < PS; ++k) s_thr[threadIdx.x * PS + k] = cnt_th[k];
            }
            __syncthreads();
        }
        // two threads per row: each keeps its half row (BN/2 values) in registers and sweeps it once per positive of the row
        {
            static_assert(GEMM_THREADS == 2 * GEMM_BM, "two threads per tile row");
            constexpr int HW = BN / 2;
            const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
            const int m = m0 + row, nb = n0 + half * HW;
            float v[HW];
#pragma unroll
            for (int j = 0; j < HW / 4; ++j) {
                const float4 t4 = *reinterpret_cast<const float4*>(Cs + row * CLD + half * HW + 4 * j);
                v[4 * j] = t4.x; v[4 * j + 1] = t4.y; v[4 * j + 2] = t4.z; v[4 * j + 3] = t4.w;
            }
            // columns past N never count: push them to the losing side of every comparison
            const float lose = (E.cnt_mode == 1) ? -INFINITY : INFINITY;
#pragma unroll
            for (int i = 0; i < HW; ++i) {
                const float x = E.apply_exp ? __expf(v[i]) : v[i];
                v[i] = (nb + i < N) ? x : lose;
            }
            int pb, np;
            if constexpr (PS > 0) { pb = s_pb[row]; np = s_np[row]; }
            else { pb = (m < M) ? E.cnt_off[m] : 0; np = (m < M) ? E.cnt_off[m + 1] - pb : 0; }
            for (int k = 0; k < np; ++k) {                       // the two threads of a row share the trip count
                const float th = (k < PS) ? s_thr[row * PS + k] : E.cnt_thr[pb + k];
                int c = 0;
                if (E.cnt_mode == 1) {
#pragma unroll
                    for (int i = 0; i < HW; ++i) c += (v[i] > th) ? 1 : 0;
                } else {
#pragma unroll
                    for (int i = 0; i < HW; ++i) c += (v[i] < th) ? 1 : 0;
                }
                c += __shfl_xor(c, 1, 64);
                if (half == 0 && c != 0) atomicAdd(E.cnt_out + pb + k, c);
            }
        }
        return;
    }
    float* cbase = E.c + (long long)zslice * E.split_stride;
    const bool vec_main = ((E.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(cbase) & 15) == 0);
    const bool vec_c2 = ((E.ldc2 & 3) == 0) && ((reinterpret_cast<uintptr_t>(E.c2) & 15) == 0) && ((E.cols_main & 3) == 0);
    const bool vec_act = (E.act_on == 0) || (((E.ld_act & 3) == 0) && ((reinterpret_cast<uintptr_t>(E.act_src) & 15) == 0));
    constexpr int NCH = GEMM_BM * C4 / GEMM_THREADS;  // chunks per thread (16 / 8)
    constexpr int UB = (NCH % 8 == 0) ? 8 : 5;        // chunks whose extras are fetched together (independent loads in flight)
    static_assert(NCH % UB == 0, "chunk batches");
    const int w1max = E.mask_on ? E.mask_ld - 1 : 0;
    for (int cb = 0; cb < NCH; cb += UB) {
        float v[UB][4], av[UB][4];
        unsigned mw0[UB], mw1[UB];
        bool fast[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int idx = threadIdx.x + (cb + u) * GEMM_THREADS;
            const int row = idx / C4, c4 = idx % C4;
            const int m = m0 + row, n = n0 + c4 * 4;
            const bool all_main = n + 3 < E.cols_main, all_c2 = n >= E.cols_main;
            fast[u] = (m < M) && (n + 3 < N) && ((all_main && vec_main && vec_act) || (all_c2 && vec_c2));
            const float4 t = *reinterpret_cast<const float4*>(Cs + row * CLD + c4 * 4);
            v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
            const int mm = fast[u] ? m : 0, nn = fast[u] ? n : 0;          // clamped: the extras' loads stay unconditional
            if (E.mask_on) {                                               // kernel-uniform
                const int cm = nn + E.mask_col0;
                mw0[u] = E.mask[(long long)mm * E.mask_ld + (cm >> 5)];
                mw1[u] = E.mask[(long long)mm * E.mask_ld + min((cm + 3) >> 5, w1max)];
            }
            if (E.act_on != 0 && vec_act) {
                const float4 a4 = *reinterpret_cast<const float4*>(E.act_src + (long long)mm * E.ld_act + (all_main ? nn : 0));
                av[u][0] = a4.x; av[u][1] = a4.y; av[u][2] = a4.z; av[u][3] = a4.w;
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int idx = threadIdx.x + (cb + u) * GEMM_THREADS;
            const int row = idx / C4, c4 = idx % C4;
            const int m = m0 + row, n = n0 + c4 * 4;
            if (fast[u]) {
                const bool all_main = n + 3 < E.cols_main;
                float g[4] = {E.drop_scale, E.drop_scale, E.drop_scale, E.drop_scale};
                if (E.mask_on) {
                    const int cm = n + E.mask_col0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = cm + q;
                        const unsigned wd = ((c >> 5) == (cm >> 5)) ? mw0[u] : mw1[u];
                        g[q] = ((wd >> (c & 31)) & 1u) ? E.drop_scale : 0.f;
                    }
                }
                if (E.act_on != 0 && all_main) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q] *= (av[u][q] > 0.f) ? 1.f : E.act_slope;
                }
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = v[u][q] * g[q];
                    o[q] = E.apply_exp ? __expf(x) : x;
                }
                float* dst = all_main ? (cbase + (long long)m * E.ldc + n) : (E.c2 + (long long)m * E.ldc2 + (n - E.cols_main));
                *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            } else if (m < M) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < N) epi_store_one(E, m, n + q, v[u][q], cbase);
            }
        }
    }
}

template <bool AK, bool BKC, int VA, int VB, int BN>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(VMat A, VMat B, const Epi E, const int M, const int N,
                                                                const int K, const int ksplit, const Tail T) {
    using GA = StageGeom<AK, VA, GEMM_BM>;
    using GB = StageGeom<BKC, VB, BN>;
    constexpr int ASZ = GA::LDS, BSZ = GB::LDS;
    constexpr int MI = (BN == 128) ? 2 : 1;          // 32-row MFMA tiles per wave along m
    constexpr int NJ = (BN == 160) ? 5 : 2;          // along n   (BN = 160: 4 x 1 waves of 32 x 160 -- a 320-column output in two tiles)
    constexpr int CLD = BN + 4;                       // row stride of the C tile staged for the epilogue
    constexpr int SMEM = (BN == 160 || 2 * (ASZ + BSZ) > GEMM_BM * CLD) ? 2 * (ASZ + BSZ) : GEMM_BM * CLD;   // (BN = 160 stores from registers)
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* As = smem;
    float* Bs = smem + 2 * ASZ;

    const int nbn = (N + BN - 1) / BN;
    // XCD-contiguous order over (k-slice, tile): workgroup b runs on XCD b % 8; after the remap each XCD owns one
    // contiguous range of (slice, row panel, column) triples, so a k-slice of both operands (split-K) / a row panel of A
    // and all of B (no split) is pulled into ONE L2 instead of all eight.
    int zslice, lb, kbeg, kend, tail_slot = -1;
    if (T.S > 0) {                                   // tail splitting (never combined with regular split-K)
        const int bid = blockIdx.x;
        if (bid < T.nfull) {
            lb = xcd_remap(bid, T.nfull); zslice = 0; kbeg = 0; kend = K;
        } else {
            tail_slot = bid - T.nfull;
            lb = T.nfull + tail_slot / T.S;
            zslice = 0;
            kbeg = (tail_slot % T.S) * T.ksplit;
            kend = min(K, kbeg + T.ksplit);
        }
    } else {
        const int ntiles = gridDim.x;
        const int vb = xcd_remap(blockIdx.x + ntiles * blockIdx.y, ntiles * gridDim.y);
        zslice = vb / ntiles; lb = vb % ntiles;
        kbeg = zslice * ksplit;
        kend = min(K, kbeg + ksplit);
    }
    // tile order: consecutive workgroups (same XCD, same L2) share the operand panel of the SHORTER grid dimension's neighbour:
    // column tiles fastest when there are more row panels than column tiles, else row panels fastest -- the long operand is then
    // streamed from HBM once instead of once per tile of the short dimension (scoring: U read 1x instead of 8x per query block)
    const int nbm = (M + GEMM_BM - 1) / GEMM_BM;
    lb += T.tile0;
    const int tm = T.row_fast ? lb % nbm : lb / nbn, tn = T.row_fast ? lb / nbm : lb % nbn;
    const int m0 = tm * GEMM_BM, n0 = tn * BN;

    // count mode: this tile's rows' positive ranges and first thresholds are fetched NOW (two dependent loads), so that their
    // latency hides under the whole k-loop instead of sitting in the epilogue
    if (E.cnt_mode == 3) {                           // pick mode: a tile without a (row, own column) pair has nothing to do (block-uniform)
        const int lo = E.cnt_off[m0], hi = E.cnt_off[min(m0 + GEMM_BM, M)];
        if (n0 >= hi || n0 + BN <= lo) return;
    }
    int cnt_pb = 0, cnt_np = 0;
    float cnt_th[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((E.cnt_mode == 1 || E.cnt_mode == 2) && threadIdx.x < GEMM_BM && m0 + (int)threadIdx.x < M) {
        cnt_pb = E.cnt_off[m0 + threadIdx.x];
        cnt_np = E.cnt_off[m0 + threadIdx.x + 1] - cnt_pb;
#pragma unroll
        for (int k = 0; k < 8; ++k) cnt_th[k] = E.cnt_thr[(k < cnt_np) ? cnt_pb + k : 0];
    }

    // clip the reduction range into the operands' own bounds (split-K and K tails read zeros)
    if (AK) { A.cols = min(A.cols, kend); A.cols_main = min(A.cols_main, kend); }
    else    { A.rows = min(A.rows, kend); }
    if (BKC) { B.cols = min(B.cols, kend); B.cols_main = min(B.cols_main, kend); }
    else     { B.rows = min(B.rows, kend); }

    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wr = (BN == 128) ? (w >> 1) : w, wc = (BN == 128) ? (w & 1) : 0;
    const int wm0 = wr * (MI * 32), wn0 = wc * 64;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[GA::NREG], rb[GB::NREG];
    unsigned ma[GA::PASSES], mb[GB::PASSES];
    const int nk = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
    // leading k-tiles that are plain for BOTH operands.  The pipelined loop is split by the kind of the tile being
    // ISSUED (fast prefix, then generic tail) so that no branch sits between a load and its first use.
    int nkf = 0;
    if (block_is_plain<AK, GEMM_BM>(A, m0) && block_is_plain<BKC, BN>(B, n0)) {
        const int full = (kend - kbeg) / GEMM_BK;     // complete k-tiles (a ragged last tile takes the generic path)
        const int fa_ = AK ? max(0, (A.cols_main - kbeg) / GEMM_BK) : min(full, max(0, (min(A.rows, A.rows_main) - kbeg) / GEMM_BK));
        const int fb_ = BKC ? max(0, (B.cols_main - kbeg) / GEMM_BK) : min(full, max(0, (min(B.rows, B.rows_main) - kbeg) / GEMM_BK));
        nkf = min(nk, min(fa_, fb_));
    }
    FastPtr<AK, VA, GEMM_BM> fpa;
    FastPtr<BKC, VB, BN> fpb;
    fast_init<AK, VA, GEMM_BM>(A, m0, kbeg, fpa);
    fast_init<BKC, VB, BN>(B, n0, kbeg, fpb);

#define TXE_COMPUTE_TILE(cur_)                                                                                       \
    {                                                                                                                \
        const float* a_l = As + (cur_) * ASZ;                                                                        \
        const float* b_l = Bs + (cur_) * BSZ;                                                                        \
        _Pragma("unroll") for (int kb = 0; kb < GEMM_BK / 8; ++kb) {                                                 \
            float fa[MI][4], fb[NJ][4];                                                                              \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) frag_load<AK, GEMM_BM>(a_l, wm0 + i * 32, kb, fa[i]);     \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) frag_load<BKC, BN>(b_l, wn0 + j * 32, kb, fb[j]);         \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                            \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                       \
                    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                   \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);    \
        }                                                                                                            \
    }
#define TXE_NOTHING
#define TXE_STAGE_GENERIC(k0_, buf_, COMPUTE_)                                                                       \
    {                                                                                                                \
        stage_issue<AK, VA, GEMM_BM, false>(A, m0, (k0_), ra, ma);                                                   \
        stage_issue<BKC, VB, BN, false>(B, n0, (k0_), rb, mb);                                                       \
        __builtin_amdgcn_sched_barrier(0); /* loads first: the whole MFMA block then covers their latency */        \
        COMPUTE_                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        stage_finish<AK, VA, GEMM_BM, false>(A, m0, (k0_), ra, ma);                                                  \
        stage_finish<BKC, VB, BN, false>(B, n0, (k0_), rb, mb);                                                      \
        stage_store<AK, VA, GEMM_BM>(As + (buf_) * ASZ, ra);                                                         \
        stage_store<BKC, VB, BN>(Bs + (buf_) * BSZ, rb);                                                             \
    }
#define TXE_STAGE_FAST(k0_, buf_, COMPUTE_)                                                                          \
    {                                                                                                                \
        fast_issue<AK, VA, GEMM_BM, true>(A, fpa, ra, ma);                                                           \
        fast_issue<BKC, VB, BN, true>(B, fpb, rb, mb);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        COMPUTE_                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        fast_finish<AK, VA, GEMM_BM>(A, fpa, ra, ma);                                                                     \
        fast_finish<BKC, VB, BN>(B, fpb, rb, mb);                                                                         \
        stage_store<AK, VA, GEMM_BM>(As + (buf_) * ASZ, ra);                                                         \
        stage_store<BKC, VB, BN>(Bs + (buf_) * BSZ, rb);                                                             \
    }

    if (nk > 0) {
        if (nkf > 0) TXE_STAGE_FAST(kbeg, 0, TXE_NOTHING)
        else TXE_STAGE_GENERIC(kbeg, 0, TXE_NOTHING)
    }
    __syncthreads();
    int t = 1;
    for (; t < nkf; ++t) {                      // tile t (plain) is fetched while tile t-1 is multiplied
        TXE_STAGE_FAST(kbeg + t * GEMM_BK, t & 1, TXE_COMPUTE_TILE((t - 1) & 1))
        __syncthreads();
    }
    for (; t < nk; ++t) {                       // generic tiles (extension columns / ragged edges)
        TXE_STAGE_GENERIC(kbeg + t * GEMM_BK, t & 1, TXE_COMPUTE_TILE((t - 1) & 1))
        __syncthreads();
    }
    if (nk > 0) TXE_COMPUTE_TILE((nk - 1) & 1)
#undef TXE_STAGE_FAST
#undef TXE_STAGE_GENERIC
#undef TXE_NOTHING
#undef TXE_COMPUTE_TILE

    if constexpr (BN == 160) {
        // plain epilogue straight from the accumulators (the 164-float C rows of a 128 x 160 tile do not fit beside nothing in the
        // operand stages' 72 KB; the launcher sends only plain / exp epilogues without tail splitting here): register e of a
        // 32 x 32 block is row (e&3) + 8*(e>>2) + 4*(lane>>5), column lane&31 -- the 32 lanes of a half wave write 128 contiguous bytes
        float* cb = E.c + (long long)zslice * E.split_stride;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + j * 32 + (l & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + 4 * (l >> 5) + (e & 3) + 8 * (e >> 2);
                const float x = acc[0][j][e];
                if (m < M && n < N) cb[(long long)m * E.ldc + n] = E.apply_exp ? __expf(x) : x;
            }
        }
        return;
    }
    // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -- a lane's registers
    // walk DOWN a column, so storing them directly issues 64 scattered dword stores per thread (store-issue bound: ~10 us of
    // a 60 us K=320 round).  The tile is transposed through the (now idle) operand stages instead: every thread then owns
    // 4 consecutive columns of a row -> one 16-byte load per extra (activation source), one or two mask words, one 16-byte
    // store, rows written as full 512-byte lines.
    float* Cs = smem;
    __syncthreads();                                 // every wave is done reading the operand stages
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm0 + i * 32 + 4 * (l >> 5) + (e & 3) + 8 * (e >> 2);
                Cs[row * CLD + wn0 + j * 32 + (l & 31)] = acc[i][j][e];
            }
    __syncthreads();
    constexpr int C4 = BN / 4;                       // 16-byte chunks per tile row
    if (tail_slot >= 0) {                            // leftover-tile slice: park the raw partial tile, fix-up kernel finishes
        float* part = T.ws + (long long)tail_slot * (GEMM_BM * BN);
        for (int idx = threadIdx.x; idx < GEMM_BM * C4; idx += GEMM_THREADS) {
            const int row = idx / C4, c4 = idx % C4;
            *reinterpret_cast<float4*>(part + row * BN + c4 * 4) = *reinterpret_cast<const float4*>(Cs + row * CLD + c4 * 4);
        }
        return;
    }
    gemm_tile_epilogue<BN, SMEM>(E, smem, m0, n0, M, N, tn, nbn, zslice, cnt_pb, cnt_np, cnt_th);
}

// ---- whole rounds of a plain product: persistent workgroups, the C tile drained under the NEXT tile's k-loop ---------------------
// A short reduction (K = 256 .. 320: the first layer's projection, the table projection, the scoring product) spends ~20 % of every
// round of 2 x 256 tiles storing C: all workgroups finish together, 33 MB leave the chip at the HBM write rate while the matrix pipe
// idles, and the next round pays a launch + first-load prologue.  Here one workgroup per slot walks tiles b, b + G, ...; when a tile's
// k-loop ends its accumulators move to a second register set and are written -- straight from registers: the 32 lanes of a half wave
// hold 32 consecutive floats of one C row, 128 bytes per store -- in four batches of 16 stores inside the first four k-tiles of the
// NEXT tile, i.e. under its MFMA blocks.  No LDS transposition, no epilogue extras: plain (or exp) epilogues, plain operands
// (optionally a dropout mask on either), 128 x 128 tiles that lie wholly inside both operands, K a multiple of 32 and >= 160.
// Accumulation order per element is the k order of gemm_kernel: results are bit-identical.
#ifndef TXE_PERSIST_MAXK
#define TXE_PERSIST_MAXK 512
#endif
struct Persist {
    float* c;
    long long ldc;
    int row_fast;
    int apply_exp;
    float* dummy;    // >= gridDim.x * 256 floats: where the lanes of a tile's rows / columns past M / N put their stores
    // work items: [0, ntile_items) whole tiles (tile = xcd_remap(item)), then n_slices k-slices of the leftover tiles -- slice i is
    // k-tiles [ (i % S) * kslice, ... ) of tile ntile_items + i / S and parks its raw partial tile at part + i * 128 * 128
    int ntile_items, S, kslice;
    float* part;
};

template <bool AK, bool BKC, int DK /* k-tiles over which a tile's 64 stores per lane are spread: 4 or 8 */,
          int LKB = GEMM_BK / 8 /* eight-column groups of the product's LAST k-tile that hold data (Epi.k_valid) */>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_persist_kernel(const VMat A, const VMat B, const Persist P, const int M, const int N,
                                                                        const int K, const int nitems) {
    constexpr int BN = 128, VA = 4, VB = 4, MI = 2, NJ = 2;
    using GA = StageGeom<AK, VA, GEMM_BM>;
    using GB = StageGeom<BKC, VB, BN>;
    constexpr int ASZ = GA::LDS, BSZ = GB::LDS;
    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];
    float* As = smem;
    float* Bs = smem + 2 * ASZ;
    const int nbn = (N + BN - 1) / BN, nbm = (M + GEMM_BM - 1) / GEMM_BM;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wm0 = (w >> 1) * 64, wn0 = (w & 1) * 64;
    const int nk_all = K / GEMM_BK;
    float* const dummy = P.dummy + (long long)blockIdx.x * GEMM_THREADS + threadIdx.x;

    f32x16 acc[MI][NJ], prev[MI][NJ];
    float* pbase = dummy;                             // previous item: address of this lane's element (row 4*(l>>5), col l&31) of its wave's
    long long pld = 0;                                // 64 x 64 block, its row pitch; rows / columns left before M / N from there
    int prem_m = 0, prem_n = 0, pexp = 0;
    bool have_prev = false;
    float ra[GA::NREG], rb[GB::NREG];
    unsigned ma[GA::PASSES], mb[GB::PASSES];

#define TXE_P_COMPUTE(cur_) TXE_P_COMPUTE_KB(cur_, GEMM_BK / 8)
#define TXE_P_COMPUTE_KB(cur_, nkb_)                                                                                 \
    {                                                                                                                \
        const float* a_l = As + (cur_) * ASZ;                                                                        \
        const float* b_l = Bs + (cur_) * BSZ;                                                                        \
        _Pragma("unroll") for (int kb = 0; kb < (nkb_); ++kb) {                                                      \
            float fa[MI][4], fb[NJ][4];                                                                              \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) frag_load<AK, GEMM_BM>(a_l, wm0 + i * 32, kb, fa[i]);     \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) frag_load<BKC, BN>(b_l, wn0 + j * 32, kb, fb[j]);         \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                            \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                       \
                    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                   \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);    \
        }                                                                                                            \
    }
    // stores of the previous item's 32 x 32 block (i_, j_) of this wave: register e is row (e&3) + 8*(e>>2) (+ 4 for the upper
    // half wave, folded into pbase), column = lane & 31.  Lanes past M / N write to their own dummy word instead (no branch).
#define TXE_P_DRAIN(i_, j_) TXE_P_DRAIN_E(i_, j_, 0, 16)
#define TXE_P_DRAIN_E(i_, j_, e0_, e1_)                                                                              \
    {                                                                                                                \
        float* p0 = pbase + (long long)((i_) * 32) * pld + (j_) * 32;                                                \
        const bool cok = (j_) * 32 < prem_n;                                                                         \
        _Pragma("unroll") for (int e = (e0_); e < (e1_); ++e) {                                                      \
            const int roff = (e & 3) + 8 * (e >> 2);                                                                 \
            const bool ok = cok & ((i_) * 32 + roff < prem_m);                                                       \
            const float x = prev[i_][j_][e];                                                                         \
            float* dst = ok ? (p0 + (long long)roff * pld) : dummy;                                                  \
            *dst = pexp ? __expf(x) : x;                                                                             \
        }                                                                                                            \
    }
#define TXE_P_NODRAIN
#define TXE_P_STAGE(buf_, COMPUTE_, DRAIN_)                                                                          \
    {                                                                                                                \
        fast_issue<AK, VA, GEMM_BM, true>(A, fpa, ra, ma);                                                           \
        fast_issue<BKC, VB, BN, true>(B, fpb, rb, mb);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        COMPUTE_                                                                                                     \
        DRAIN_                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        fast_finish<AK, VA, GEMM_BM>(A, fpa, ra, ma);                                                                \
        fast_finish<BKC, VB, BN>(B, fpb, rb, mb);                                                                    \
        stage_store<AK, VA, GEMM_BM>(As + (buf_) * ASZ, ra);                                                         \
        stage_store<BKC, VB, BN>(Bs + (buf_) * BSZ, rb);                                                             \
    }

    // The k-tiles of a workgroup's items form ONE software-pipelined stream: k-tile t+1 is fetched while k-tile t is multiplied, and
    // the first k-tile of the NEXT item is fetched under the last MFMA block of the current one.  `par` is the stage-buffer parity
    // of the current item's k-tile 0.
    int par = 0;
    int m0 = 0, n0 = 0, nki = 0, slice = -1;          // located item: tile origin, k-tiles, slice index (-1: a whole tile)
    bool kend = true;                                 // ... and whether its k range ends at K (a whole tile, or a tile's last slice)
    FastPtr<AK, VA, GEMM_BM> fpa;
    FastPtr<BKC, VB, BN> fpb;
    auto locate = [&](const int item) {
        int lb, kb0;
        if (DK == 8 || item < P.ntile_items) { lb = xcd_remap(item, P.ntile_items); kb0 = 0; nki = nk_all; slice = -1; kend = true; }
        else {                                        // (k-slices only with the 4-step drain schedule: a slice may be 5 k-tiles short)
            slice = item - P.ntile_items;
            lb = P.ntile_items + slice / P.S;
            kb0 = (slice % P.S) * P.kslice;
            nki = min(P.kslice, nk_all - kb0);
            kend = kb0 + nki == nk_all;
        }
        const int tm = P.row_fast ? lb % nbm : lb / nbn, tn = P.row_fast ? lb / nbm : lb % nbn;
        m0 = tm * GEMM_BM; n0 = tn * BN;
        fast_init<AK, VA, GEMM_BM>(A, m0, kb0 * GEMM_BK, fpa);
        fast_init<BKC, VB, BN>(B, n0, kb0 * GEMM_BK, fpb);
    };
    int item = blockIdx.x;
    if (item < nitems) {
        locate(item);
        TXE_P_STAGE(0, TXE_P_NODRAIN, TXE_P_NODRAIN)
        __syncthreads();
    }
#define TXE_P_STEP(c_, DRAIN_)                                                                                       \
    TXE_P_STAGE(((c_) + 1 + par) & 1, TXE_P_COMPUTE(((c_) + par) & 1), DRAIN_)                                       \
    __syncthreads();
    while (item < nitems) {
        const int cm0 = m0, cn0 = n0, cnk = nki, cslice = slice;
        const bool ckend = kend;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        int t = 0;
        if (have_prev) {                              // (uniform; every load is consumed on the side of the branch that issued it)
            if constexpr (DK == 8) {
                TXE_P_STEP(0, TXE_P_DRAIN_E(0, 0, 0, 8))
                TXE_P_STEP(1, TXE_P_DRAIN_E(0, 0, 8, 16))
                TXE_P_STEP(2, TXE_P_DRAIN_E(0, 1, 0, 8))
                TXE_P_STEP(3, TXE_P_DRAIN_E(0, 1, 8, 16))
                TXE_P_STEP(4, TXE_P_DRAIN_E(1, 0, 0, 8))
                TXE_P_STEP(5, TXE_P_DRAIN_E(1, 0, 8, 16))
                TXE_P_STEP(6, TXE_P_DRAIN_E(1, 1, 0, 8))
                TXE_P_STEP(7, TXE_P_DRAIN_E(1, 1, 8, 16))
                t = 8;
            } else {                                  // (every item has at least 5 k-tiles)
                TXE_P_STEP(0, TXE_P_DRAIN(0, 0))
                TXE_P_STEP(1, TXE_P_DRAIN(0, 1))
                TXE_P_STEP(2, TXE_P_DRAIN(1, 0))
                TXE_P_STEP(3, TXE_P_DRAIN(1, 1))
                t = 4;
            }
        }
        for (; t < cnk - 1; ++t) { TXE_P_STEP(t, TXE_P_NODRAIN) }
        const int next = item + (int)gridDim.x;
        if (next < nitems) {                          // last k-tile: the loads in flight are the next item's first k-tile
            locate(next);
            if constexpr (LKB < GEMM_BK / 8) {        // (uniform branch; each side issues and consumes its own loads)
                if (ckend) { TXE_P_STAGE((cnk + par) & 1, TXE_P_COMPUTE_KB((cnk - 1 + par) & 1, LKB), TXE_P_NODRAIN) __syncthreads(); }
                else { TXE_P_STEP(cnk - 1, TXE_P_NODRAIN) }
            } else {
                TXE_P_STEP(cnk - 1, TXE_P_NODRAIN)
            }
        } else {
            if constexpr (LKB < GEMM_BK / 8) {
                if (ckend) TXE_P_COMPUTE_KB((cnk - 1 + par) & 1, LKB)
                else TXE_P_COMPUTE((cnk - 1 + par) & 1)
            } else {
                TXE_P_COMPUTE((cnk - 1 + par) & 1)
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) prev[i][j] = acc[i][j];
        if (DK == 8 || cslice < 0) {
            const int r0 = cm0 + wm0 + 4 * (l >> 5), c0 = cn0 + wn0 + (l & 31);
            pbase = P.c + (long long)r0 * P.ldc + c0;
            pld = P.ldc; prem_m = M - r0; prem_n = N - c0; pexp = P.apply_exp;
        } else {                                      // a k-slice parks its raw partial tile; gemm_tail_fixup_kernel finishes
            pbase = P.part + (long long)cslice * (GEMM_BM * BN) + (wm0 + 4 * (l >> 5)) * BN + wn0 + (l & 31);
            pld = BN; prem_m = GEMM_BM; prem_n = BN; pexp = 0;
        }
        have_prev = true;
        par = (par + cnk) & 1;
        item = next;
    }
#undef TXE_P_STEP
    if (have_prev) {
        TXE_P_DRAIN(0, 0)
        TXE_P_DRAIN(0, 1)
        TXE_P_DRAIN(1, 0)
        TXE_P_DRAIN(1, 1)
    }
#undef TXE_P_STAGE
#undef TXE_P_NODRAIN
#undef TXE_P_DRAIN
#undef TXE_P_DRAIN_E
#undef TXE_P_COMPUTE
#undef TXE_P_COMPUTE_KB
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

int device_cu_count();     // txe_profile.hip (cached hipDeviceAttributeMultiprocessorCount of the current device)

// 128 x BN tile choice: the narrower tile when it wastes fewer MFMA columns or needs fewer (fractional) rounds of
// workgroups over the CUs (2 co-resident workgroups per CU).
// split-K products take 160-wide tiles only when they are big: their 512 fat workgroups (256 VGPRs) leave a concurrent kernel of the
// other stream no wave slot -- the 8.5-GFLOP folded-layer weight gradient slowed the sweep running beside it by 30 us and the step
// by 10, while the 153-GFLOP products of the 2-layer model gain 2 % from the 4.4 % fewer padded columns
// ... above 20 GFLOP, and also where 64-wide tiles pad no more columns (N = 320 = 2 x 160 = 5 x 64: the first layer's weight
// gradient -- 32 tiles x 16 slices fill the 512 slots exactly and run the full-rate 4-wave k-loop: 219 -> 205 us.  While that
// product shared the chip with the skinny d_X GEMM on the second stream the fat workgroups cost more than they gained; d_X is now an
// in-line HBM stream, txe_dxpos.hip.)
constexpr double BN160_SPLIT_MIN_FLOPS = 2e10;

static inline int choose_bn(int M, int N, int splits, bool tail_split = false, int K = 0, bool allow160 = false, bool force160 = false) {
    if (allow160 && splits > 1 && force160) return 160;
    if (N <= 64) return 64;
    const int slots = 2 * device_cu_count();
    if (allow160 && splits > 1 && 2.0 * M * (double)N * K >= BN160_SPLIT_MIN_FLOPS) {
        // split-K products (weight gradients): 128 x 160 tiles when they cover N with fewer padded columns than 128-wide ones and at
        // least as few as 64-wide ones -- 320 = 2 x 160 runs the full-rate 4-wave k-loop where 5 x 64 starves the matrix pipe
        const int w160 = ((N + 159) / 160) * 160, w128 = ((N + 127) / 128) * 128, w64 = ((N + 63) / 64) * 64;
        if (w160 < w128 && w160 <= w64) return 160;
    }
    if (allow160 && splits == 1) {
        // one round of 160-wide tiles where 128-wide ones spill into a second, k-split round with its fix-up launch
        // (dZ = d_hg W: 4096 x 2080 is 32 x 13 = 416 tiles of 160 columns, 544 of 128)
        const int nbm = (M + GEMM_BM - 1) / GEMM_BM;
        const int t160 = nbm * ((N + 159) / 160), t128 = nbm * ((N + 127) / 128);
        const int w160 = ((N + 159) / 160) * 160, w128 = ((N + 127) / 128) * 128;
        if (t160 <= slots && t128 > slots && t160 * 5 >= slots * 3 && w160 <= w128) return 160;
    }
    auto cost = [&](int bn) {
        const long long blocks = (long long)((M + GEMM_BM - 1) / GEMM_BM) * ((N + bn - 1) / bn) * splits;
        const long long rounds = (blocks + slots - 1) / slots;
        double r = (double)rounds;
        // with tail splitting a partial last round is cut along k: its time shrinks to its share of the slots (down to 4 k-tiles)
        if (tail_split && splits == 1 && blocks % slots != 0) {
            const double frac = (double)(blocks % slots) / slots;
            const double kfloor = K > 0 ? 4.0 * GEMM_BK / K : 0.25;
            r = (double)(blocks / slots) + (frac > kfloor ? frac : kfloor) + 0.1;      // + fix-up kernel
        }
        // per-flop cost of the narrow tile, measured: x1.6 on the forward / dX shapes (short k-loops, epilogue extras), x1.1 on the
        // long split-K reductions of the weight gradients (2.5 vs 4.7 us per k-tile and round)
        return r * bn * (bn == 64 ? (splits > 1 ? 1.1 : 1.6) : 1.0);
    };
    return cost(64) < cost(128) ? 64 : 128;
}

template <bool AK, bool BKC, int VA, int VB>
static inline void gemm_launch_v(int bn, dim3 grid, hipStream_t stream, const VMat& A, const VMat& B, const Epi& E, int M, int N,
                                 int K, int ksplit, const Tail& T) {
    // profiler record named exactly like rocprofv3 prints the kernel (minus "void txe::"), so that bench.py can join its
    // HIP-event timings with the committed rocprof summaries under profiles/
    char* name = nullptr;
    static char names[3][64];
    static bool init = false;
    if (!init) {
        for (int b = 0; b < 3; ++b)
            snprintf(names[b], sizeof(names[b]), "gemm_kernel<%s, %s, %d, %d, %d>", AK ? "true" : "false", BKC ? "true" : "false", VA, VB,
                     b == 0 ? 128 : (b == 1 ? 64 : 160));
        init = true;
    }
    name = names[bn == 128 ? 0 : (bn == 64 ? 1 : 2)];
    ProfScope prof(name, stream, E.alg_flops > 0.0 ? E.alg_flops : 2.0 * M * (double)N * K, 0);
    if constexpr (!BKC && VA == 4 && VB == 4) {      // (160-wide tiles: B row-contiguous -- weight gradients (TN), dZ = d_hg W (NN))
        if (bn == 160) {
            hipLaunchKernelGGL((gemm_kernel<AK, BKC, VA, VB, 160>), grid, dim3(GEMM_THREADS), 0, stream, A, B, E, M, N, K, ksplit, T);
            return;
        }
    }
    if (bn == 128) hipLaunchKernelGGL((gemm_kernel<AK, BKC, VA, VB, 128>), grid, dim3(GEMM_THREADS), 0, stream, A, B, E, M, N, K, ksplit, T);
    else hipLaunchKernelGGL((gemm_kernel<AK, BKC, VA, VB, 64>), grid, dim3(GEMM_THREADS), 0, stream, A, B, E, M, N, K, ksplit, T);
}

// Launch C = A*B.  splits > 1 => split-K over gridDim.y, block z stores at E.c + z*E.split_stride.
// Vector widths: 16-byte loads for an operand whose rows are 16-byte aligned, else 8-byte; a 4-byte-only operand
// drops both to scalar loads (odd leading dimensions: correctness path, not a fast one).
// bytes of tail-splitting workspace that always suffice (r*S <= slots partial tiles of 128x128 floats)
static inline size_t gemm_tail_ws_bytes() {       // (+ the persistent kernel's dummy words in front of its slices' partial tiles)
    return (size_t)2 * device_cu_count() * (GEMM_BM * 128 + GEMM_THREADS) * sizeof(float);
}

// txe_gemm_tn.hip: the split-K TN product on 128 x 160 tiles with LDS-direct operand copies (txe_gemm_tnlds.h); TXE_ERR_ARG = not
// eligible (the caller falls through to gemm_kernel)
int gemm_tn_lds_launch(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, int ksplit, hipStream_t stream);

template <bool AK, bool BKC>
static inline int gemm_launch_layout(const VMat& A, const VMat& B, const Epi& E_in, int M, int N, int K, int splits,
                                     hipStream_t stream, void* tail_ws = nullptr, size_t tail_ws_bytes = 0) {
    if (M <= 0 || N <= 0) return TXE_OK;
    Epi E = E_in;
    {   // the epilogue loads its extras unconditionally: give the unused ones a readable dummy address
        const void* valid = E.c ? (const void*)E.c : (const void*)E.c2;
        if (!E.act_on) E.act_src = (const float*)valid;
        if (!E.mask_on) E.mask = (const unsigned*)valid;
    }
    int va = vmat_vec(A), vb = vmat_vec(B);
    if (va == 1 || vb == 1) va = vb = 1;
    if (splits < 1) splits = 1;
    const bool allow160 = !BKC && va == 4 && vb == 4 && !E.mask_on && !E.act_on && E.cnt_mode == 0 &&
                          (E.c2 == nullptr || E.cols_main >= N);
    const bool tail_split = tail_ws != nullptr && !E.plain_k_order;
    int bn = E.force_bn128 ? 128 : choose_bn(M, N, splits, tail_split, K, allow160, (E.route & GEMM_ROUTE_FORCE_BN160) != 0);
    int tile0 = 0;
    // whole rounds of 128 x 128 tiles of a plain product with a short reduction: persistent workgroups (gemm_persist_kernel)
    if (splits == 1 && tail_ws != nullptr && va == 4 && vb == 4 && N > 64 && !(E.route & GEMM_ROUTE_NO_PERSIST) && !E.mask_on && !E.act_on &&
        E.cnt_mode == 0 && (E.c2 == nullptr || E.cols_main >= N) && (K % GEMM_BK) == 0 && K >= 5 * GEMM_BK && K <= TXE_PERSIST_MAXK &&
        A.p2 == nullptr && B.p2 == nullptr && A.cols_main == A.cols && B.cols_main == B.cols && A.rows_main >= A.rows && B.rows_main >= B.rows &&
        (AK ? A.cols : A.rows) >= K && (BKC ? B.cols : B.rows) >= K) {
        const int slots = 2 * device_cu_count();
        const int nbm = (M + GEMM_BM - 1) / GEMM_BM, nbn = (N + 127) / 128;
        const int row_fast = (nbn > nbm) ? 1 : 0;
        // (ragged last panels are fine: k-contiguous operands re-read a valid row for the passes past the end, row-contiguous ones
        //  zero the column vectors past it)
        const int ntiles = nbm * nbn, nfull = (ntiles / slots) * slots, r = ntiles - nfull;
        const int nkt = K / GEMM_BK;
        const size_t dummy_bytes = (size_t)slots * GEMM_THREADS * sizeof(float);
        if (nfull >= 2 * slots && tail_ws_bytes >= dummy_bytes) {
            Persist P;
            P.c = E.c; P.ldc = E.ldc; P.row_fast = row_fast; P.apply_exp = E.apply_exp; P.dummy = (float*)tail_ws;
            P.ntile_items = nfull; P.S = 1; P.kslice = nkt; P.part = nullptr;
            // the leftover tiles (a last partial round) as k-slices INSIDE the same launch: every workgroup's last item is then a
            // slice of >= 5 k-tiles whose k-loop still drains the tile before it; a fix-up launch adds the slices in fixed order
            int S = (r > 0 && tail_split) ? slots / r : 0;
            if (S > nkt / 5) S = nkt / 5;
            if (S > 16) S = 16;
            int kslice = S >= 2 ? (nkt + S - 1) / S : nkt;
            if (S >= 2 && (nkt - (S - 1) * kslice < 5 || dummy_bytes + (size_t)r * S * GEMM_BM * 128 * sizeof(float) > tail_ws_bytes)) S = 0;
            int dk = K >= 9 * GEMM_BK ? 8 : 4;
            int nitems = nfull;
            if (S >= 2) {
                dk = 4;
                P.S = S; P.kslice = kslice; P.part = (float*)((char*)tail_ws + dummy_bytes);
                nitems = nfull + r * S;
            } else if (r > 0 && E.plain_k_order) {
                // no k-split allowed (the scoring loop compares scores of different launches bit for bit): the leftover tiles ride as
                // whole tiles on the first r workgroups -- the same makespan as a second launch of r workgroups, without the launch
                // and with their C stores drained under nothing worse than the kernel's tail (MAG-CS scoring: 297 + 69 us -> one launch)
                P.ntile_items = ntiles;
                nitems = ntiles;
            }
            // zero padding behind k_valid: the last k-tile's groups of eight columns that hold data (NT products; 2 of 4 or nothing)
            const int kv = (AK && BKC && E.k_valid > 0 && E.k_valid <= K) ? E.k_valid : K;
            const int lkb = (kv - (K - GEMM_BK) <= 16 && kv > K - GEMM_BK) ? 2 : 4;
            static char names[4][48];                    // (per template instantiation of this launcher: one AK / BKC pair)
            char* name = names[(dk == 8) + 2 * (lkb == 2)];
            if (!name[0]) snprintf(name, 48, lkb == 2 ? "gemm_persist_kernel<%s, %s, %d, 2>" : "gemm_persist_kernel<%s, %s, %d, 4>", AK ? "true" : "false", BKC ? "true" : "false", dk);
            const double all = E.alg_flops > 0.0 ? E.alg_flops : 2.0 * M * (double)N * K;
            const double share = (S >= 2 || nitems == ntiles) ? 1.0 : (double)nfull / (double)ntiles;
            {
                ProfScope prof(name, stream, all * share, 0);
                bool launched = false;
                if constexpr (AK && BKC) {
                    if (lkb == 2) {
                        if (dk == 8) hipLaunchKernelGGL((gemm_persist_kernel<AK, BKC, 8, 2>), dim3(slots), dim3(GEMM_THREADS), 0, stream, A, B, P, M, N, K, nitems);
                        else hipLaunchKernelGGL((gemm_persist_kernel<AK, BKC, 4, 2>), dim3(slots), dim3(GEMM_THREADS), 0, stream, A, B, P, M, N, K, nitems);
                        launched = true;
                    }
                }
                if (launched) {}
                else if (dk == 8) hipLaunchKernelGGL((gemm_persist_kernel<AK, BKC, 8>), dim3(slots), dim3(GEMM_THREADS), 0, stream, A, B, P, M, N, K, nitems);
                else hipLaunchKernelGGL((gemm_persist_kernel<AK, BKC, 4>), dim3(slots), dim3(GEMM_THREADS), 0, stream, A, B, P, M, N, K, nitems);
                TXE_CHECK_LAUNCH();
            }
            if (S >= 2) {
                Tail T;
                T.nfull = nfull; T.S = S; T.ksplit = kslice * GEMM_BK; T.ws = P.part; T.row_fast = row_fast; T.tile0 = 0;
                ProfScope prof("gemm_tail_fixup_kernel<128>", stream, 4.0 * r * GEMM_BM * 128 * (S + 1.0), 1);
                hipLaunchKernelGGL((gemm_tail_fixup_kernel<128>), dim3(r, 16), dim3(256), 0, stream, E, T, M, N);
                TXE_CHECK_LAUNCH();
                return TXE_OK;
            }
            if (r == 0 || nitems == ntiles) return TXE_OK;
            tile0 = nfull;
            bn = 128;                                      // (the rest keeps the persistent part's tile numbering)
            E.alg_flops = all * (1.0 - share);
        }
    }
    const int nbm = (M + GEMM_BM - 1) / GEMM_BM, nbn = (N + bn - 1) / bn;
    int ksplit = (K + splits - 1) / splits;
    ksplit = ((ksplit + GEMM_BK - 1) / GEMM_BK) * GEMM_BK;
    if (ksplit == 0) ksplit = GEMM_BK;
    dim3 grid(nbm * nbn - tile0, splits);
    Tail T;
    T.nfull = 0; T.S = 0; T.ksplit = 0; T.ws = (float*)tail_ws;
    T.row_fast = (nbn > nbm) ? 1 : 0;
    T.tile0 = tile0;
    if (splits == 1 && tail_split) {
        const int slots = 2 * device_cu_count();
        const int tiles = nbm * nbn - tile0, r = tiles % slots;
        const int nkt = (K + GEMM_BK - 1) / GEMM_BK;
        int S = (r > 0) ? slots / r : 0;
        if (S > nkt / 4) S = nkt / 4;                 // >= 4 k-tiles per slice
        if (S > 16) S = 16;
        if (S >= 2 && (size_t)r * S * GEMM_BM * bn * sizeof(float) <= tail_ws_bytes) {
            T.nfull = tiles - r; T.S = S;
            T.ksplit = ((nkt + S - 1) / S) * GEMM_BK;
            grid = dim3(T.nfull + r * S, 1);
        }
    }
    if constexpr (!AK && !BKC) {
        if (bn == 160 && splits > 1 && va == 4 && vb == 4 && tile0 == 0 && T.S == 0) {
            const int rc = gemm_tn_lds_launch(A, B, E, M, N, K, splits, ksplit, stream);
            if (rc != TXE_ERR_ARG) return rc;
        }
    }
    if (va == 4 && vb == 4) gemm_launch_v<AK, BKC, 4, 4>(bn, grid, stream, A, B, E, M, N, K, ksplit, T);
    else if (va == 4 && vb == 2) gemm_launch_v<AK, BKC, 4, 2>(bn, grid, stream, A, B, E, M, N, K, ksplit, T);
    else if (va == 2 && vb == 4) gemm_launch_v<AK, BKC, 2, 4>(bn, grid, stream, A, B, E, M, N, K, ksplit, T);
    else if (va == 2 && vb == 2) gemm_launch_v<AK, BKC, 2, 2>(bn, grid, stream, A, B, E, M, N, K, ksplit, T);
    else gemm_launch_v<AK, BKC, 1, 1>(bn, grid, stream, A, B, E, M, N, K, ksplit, T);
    TXE_CHECK_LAUNCH();
    if (T.S > 0) {
        const int r = nbm * nbn - tile0 - T.nfull;
        ProfScope prof(bn == 128 ? "gemm_tail_fixup_kernel<128>" : "gemm_tail_fixup_kernel<64>", stream,
                       4.0 * r * GEMM_BM * bn * (T.S + 1.0), 1);      // reads S partial tiles, writes one
        if (bn == 128) hipLaunchKernelGGL((gemm_tail_fixup_kernel<128>), dim3(r, 16), dim3(256), 0, stream, E, T, M, N);
        else hipLaunchKernelGGL((gemm_tail_fixup_kernel<64>), dim3(r, 16), dim3(256), 0, stream, E, T, M, N);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

// number of split-K slices for a product with `tiles` output tiles and reduction length K: fill whole rounds of
// 2 workgroups per CU, keep >= 8 k-tiles per slice.
static inline int choose_splits(int M, int N, int K) {
    const int slots = 2 * device_cu_count();
    const int max_by_k = (K + 255) / 256 > 0 ? (K + 255) / 256 : 1;
    const int nkt = (K + GEMM_BK - 1) / GEMM_BK;
    int best = 1;
    double best_cost = 1e30;
    {   // 160-wide tiles (choose_bn's rule; the weight gradients are the TN products that get them)
        const int w160 = ((N + 159) / 160) * 160, w128 = ((N + 127) / 128) * 128, w64 = ((N + 63) / 64) * 64;
        if (N > 64 && w160 < w128 && w160 <= w64 && 2.0 * M * (double)N * K >= BN160_SPLIT_MIN_FLOPS) {
            const int tiles = ((M + GEMM_BM - 1) / GEMM_BM) * (w160 / 160);
            for (int s = 2; s <= 64 && s <= max_by_k; ++s) {
                const long long blocks = (long long)tiles * s;
                const long long rounds = (blocks + slots - 1) / slots;
                const double kt = (double)((nkt + s - 1) / s);
                const double cost = (double)rounds * (kt + 2.0) * 1.25 + 0.004 * nkt * s * 1.25;
                if (cost < best_cost) { best_cost = cost; best = s; }
            }
            if (best > 1) return best;
            best_cost = 1e30;
        }
    }
    for (int bn = 64; bn <= 128; bn += 64) {                      // the launcher's choose_bn() applies the same per-tile costs
        if (bn == 64 && N <= 64) continue;
        const int tiles = ((M + GEMM_BM - 1) / GEMM_BM) * ((N + bn - 1) / bn);
        for (int s = 1; s <= 64 && s <= max_by_k; ++s) {
            const long long blocks = (long long)tiles * s;
            const long long rounds = (blocks + slots - 1) / slots;
            const double kt = (double)((nkt + s - 1) / s);
            // time ~ rounds x k-tiles per slice x per-k-tile cost of the tile width; small penalty for the partial traffic
            const double cost = (double)rounds * (kt + 2.0) * (bn == 64 ? 0.55 : 1.0) + 0.004 * nkt * s * (bn / 128.0);
            if (cost < best_cost) { best_cost = cost; best = s; }
        }
    }
    return best;
}

// each defined in its own translation unit (txe_gemm_nt.hip / _nn / _tn) so that the 30 kernel variants compile in parallel
// tail_ws (optional, >= gemm_tail_ws_bytes()) enables tail splitting when splits == 1.
int gemm_nt(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s, void* tail_ws = nullptr, size_t tail_ws_bytes = 0);  // A[m][k], B[n][k]
int gemm_nn(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s, void* tail_ws = nullptr, size_t tail_ws_bytes = 0);  // A[m][k], B[k][n]
int gemm_tn(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s, void* tail_ws = nullptr, size_t tail_ws_bytes = 0);  // A[k][m], B[k][n]

}  // namespace txe
