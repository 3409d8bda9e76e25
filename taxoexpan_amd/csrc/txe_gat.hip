// GAT message/reduce for batched egonets on gfx950: the fused replacement of
//   apply_edges(edge_attention)      model_zoo.py:90,106-109   e = leaky_relu_0.2(a1[src] + a2[dst])
//   edge_softmax over in-edges        model_zoo.py:111-112
//   attn_drop                         model_zoo.py:114
//   update_all(src_mul_edge, sum)     model_zoo.py:95           out[v] = sum_e a_drop[e] * ft[src_e]
// (+ the inter-layer F.leaky_relu of model_zoo.py:216 as an epilogue) and of its backward.
//
// Mapping: ONE 64-lane wavefront per destination node.  Neighbour lists come from a destination-sorted
// CSR (coalesced int32 reads).  The per-destination softmax is a wavefront segmented max / sum: lane p
// owns in-edge p, __shfl_xor butterflies reduce over the segment, per head.  The normalised (and
// dropped) attention of a 64-edge chunk is parked in LDS, then all 64 lanes sweep the H*D-wide feature
// row of every neighbour with 16-byte loads (lane = feature column block) and accumulate in registers;
// nothing but `out` (and alpha, kept for backward) is written.  HBM-bound: per node it reads deg rows
// and writes one row of H*D floats.  Workgroups are remapped so that one XCD (one L2) owns a contiguous
// range of destination nodes -- the nodes of an egonet share their source rows.
#include "txe_gather.h"

namespace txe {

constexpr int SLICE_NI = 2;      // feature vectors per lane of a wave that owns a quarter of a row
#ifndef TXE_SPLIT_LIGHT_DEG
#define TXE_SPLIT_LIGHT_DEG 16
#endif
constexpr bool SPLIT_HYBRID = TXE_SPLIT_LIGHT_DEG > 0;
constexpr int SPLIT_LIGHT_DEG = TXE_SPLIT_LIGHT_DEG;   // largest out-degree a workgroup's nodes may have for the wave-per-node path

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// NX: the epilogue also forms the attention logits of the NEXT GATLayer in folded form (one head): the row this wave has just
// produced is the feature part of that layer's input X' = [out | position columns | 0] (out IS X', ld_out = nx_kp), so
//   nx_a12[v][r] = nx_scale * < X'[v] * keep(nx_mask), nx_wa[r] >,  r = 0, 1
// costs two FMAs per element here instead of a second sweep over X' (txe_gat_collapse_fwd's logits kernel).
struct NextLogits {
    const float* wa;          // [2][kp] folded attention rows of the next layer
    const unsigned* mask;     // its feature-dropout keep bits [N][mask_ld] or NULL
    float* a12;               // [N][2]
    float scale;
    int kp, mask_ld;
};
// one destination node v, one wave: attention softmax over its in-edges, aggregation, (NX) the next layer's folded logits
// NX: 0 = plain aggregation, 1 = + the next (folded) layer's logits, 2 = the same with that layer's feature-dropout mask
// TAB: the projected features are rows of a TABLE (eval-mode first layer of a batch drawn from a taxonomy's feature table):
//   ft[u] = T[rid[u]] + T2[pos[u]]     (T = table x W^T, T2 = position embedding x W_p^T, attention columns included)
// formed on the fly -- same operation order as materialising the rows first (one add, then the weighted sum), so both routes agree bit
// for bit.  The few rows of T2 sit in LDS.
struct TabSrc {
    const int* rid;           // [N] table row of every batch node
    const int* pos;           // [N] row of T2 of every batch node
    const float* t2;          // [vocab][ld_ft] in global memory (copied to LDS by the kernel)
    int vocab;
};

template <int VEC, int NI, int EU>
__device__ __forceinline__ void gather_step_tab(const float* __restrict__ base, long long ld, const int* s_idx, const int* s_pos,
                                                const float* s_t2, const float* s_w, int e, int j0, int j1, const int* hidx,
                                                float (&acc)[NI][VEC]) {
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
    for (int u = 0; u < EU; ++u) {
        const float* trow = s_t2 + (long long)s_pos[e + u] * ld;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = j0 + l + 64 * i;
            const int jc = (j < j1) ? j : j0;
            float t[VEC];
            vload<VEC>(trow + (long long)jc * VEC, t);
            const float a = s_w[hidx[i] * 64 + e + u];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[i][k] = fmaf(a, v[u][i][k] + t[k], acc[i][k]);
        }
    }
}

// the head of a node's dependent-load chain (rowptr -> col -> attention terms), fetched for several nodes of a wave at once
struct NodeHead {
    int beg, end, u;          // in-edge range; this lane's source node (0 when the lane has no edge)
    float e[4];               // this lane's edge: leaky(a_src[u] + a_dst[v]) per head (-inf: no edge / no such head); only when deg <= 64, H <= 4
};

template <int VEC, int NI, int NX, bool TAB, bool PRE = false>
__device__ __forceinline__ void gat_fwd_node(const int v, const int l, float* __restrict__ s_w, int* __restrict__ s_idx,
    float* __restrict__ s_stat, const float* __restrict__ s_wa,
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ ft,
    const long long ld_ft, const float* __restrict__ a_src, const float* __restrict__ a_dst, const int ld_a, const int H,
    const int D, const float slope, const float drop_p, const float drop_scale, const unsigned long long seed,
    const int out_mode, const float act_slope, float* __restrict__ out, const long long ld_out, float* __restrict__ alpha,
    const NextLogits& nx, const TabSrc& tab, int* __restrict__ s_pos, const float* __restrict__ s_t2, const NodeHead* head = nullptr) {
    int beg, end;
    if constexpr (PRE) { beg = head->beg; end = head->end; }
    else { beg = rowptr[v]; end = rowptr[v + 1]; }
    // attention terms of a node: TAB forms them from the table rows exactly as the materialised row would hold them
    auto att = [&](const int u, const int h, const bool dst) -> float {
        const int c = H * D + (dst ? H : 0) + h;
        if constexpr (TAB) return ft[(long long)tab.rid[u] * ld_ft + c] + s_t2[(long long)tab.pos[u] * ld_ft + c];
        else return (dst ? a_dst : a_src)[(long long)u * ld_a + h];
    };

    // Common case (every egonet: in-degree <= 51, H <= 4): one edge per lane, the logits of all heads stay in registers, the
    // H max / sum butterflies run interleaved, alpha goes straight to LDS -- one dependent-load chain instead of three.
    const bool single = (end - beg <= 64) && (H <= 4);
    if (single) {
        const int p = beg + l;
        const bool valid = p < end;
        int u;
        float e[4], ex[4], m[4], sm[4];
        if constexpr (PRE) {
            u = head->u;
#pragma unroll
            for (int h = 0; h < 4; ++h) e[h] = head->e[h];
        } else {
            u = valid ? col[p] : 0;
#pragma unroll
            for (int h = 0; h < 4; ++h)
                e[h] = (valid && h < H) ? leaky(att(u, h, false) + att(v, h, true), slope) : -INFINITY;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) m[h] = wave_max(e[h]);
#pragma unroll
        for (int h = 0; h < 4; ++h) ex[h] = (valid && h < H) ? __expf(e[h] - m[h]) : 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) sm[h] = wave_sum(ex[h]);
        if (valid) {
            if constexpr (TAB) { s_idx[l] = tab.rid[u]; s_pos[l] = tab.pos[u]; }
            else s_idx[l] = u;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if (h < H) {
                    const float al = ex[h] / sm[h];
                    if (alpha != nullptr) alpha[(long long)p * H + h] = al;
                    float f = 1.f;
                    if (drop_p > 0.f) f = drop_factor(seed, (unsigned long long)p * H + h, drop_p, drop_scale);
                    s_w[h * 64 + l] = al * f;
                }
            }
        }
    } else {
    // wavefront segmented max / sum of the attention logits of v's in-edges, per head
    for (int h = 0; h < H; ++h) {
        const float ad = att(v, h, true);
        float m = -INFINITY;
        for (int p = beg + l; p < end; p += 64) m = fmaxf(m, leaky(att(col[p], h, false) + ad, slope));
        m = wave_max(m);
        float s = 0.f;
        for (int p = beg + l; p < end; p += 64) s += __expf(leaky(att(col[p], h, false) + ad, slope) - m);
        s = wave_sum(s);
        if (l == 0) { s_stat[2 * h] = m; s_stat[2 * h + 1] = 1.f / s; }
    }
    }
    __builtin_amdgcn_wave_barrier();

    const int F = H * D, nvec = F / VEC;
    constexpr int EU = NI >= 8 ? 1 : 2;
    float nx1 = 0.f, nx2 = 0.f;
    float tx[2];                                       // NX: the (at most 128) columns behind the feature part -- position embedding and
    unsigned tk[2];                                    // zero padding, written by the next layer's preparation -- fetched ahead of the gather
    if constexpr (NX == 1 || NX == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = F + l + 64 * i, cc = min(c, nx.kp - 1);
            tx[i] = out[(long long)v * ld_out + cc];
            unsigned wd = 0xFFFFFFFFu;
            if constexpr (NX == 2) wd = nx.mask[(long long)v * nx.mask_ld + (cc >> 5)];
            tk[i] = (c < nx.kp) ? ((wd >> (cc & 31)) & 1u) : 0u;
        }
    }
    for (int t0 = 0; t0 < nvec; t0 += 64 * NI) {
        int hidx[NI];
        float acc[NI][VEC];
        unsigned kb[NI];                               // NX: keep bits of this lane's vectors, fetched ahead of the gather
        if constexpr (NX) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = t0 + l + 64 * i;
                const int c = ((j < nvec) ? j : 0) * VEC;
                kb[i] = 0xFFFFFFFFu;
                if constexpr (NX >= 2) kb[i] = nx.mask[(long long)v * nx.mask_ld + (c >> 5)] >> (c & 31);
                kb[i] = (j < nvec) ? kb[i] : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t0 + l + 64 * i;
            hidx[i] = (j < nvec) ? (j * VEC) / D : 0;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
        }
        for (int cb = beg; cb < end; cb += 64) {
            const int p = cb + l;
            if (!single && p < end) {
                const int u = col[p];
                if constexpr (TAB) { s_idx[l] = tab.rid[u]; s_pos[l] = tab.pos[u]; }
                else s_idx[l] = u;
                for (int h = 0; h < H; ++h) {
                    const float e = leaky(att(u, h, false) + att(v, h, true), slope);
                    const float al = __expf(e - s_stat[2 * h]) * s_stat[2 * h + 1];
                    if (alpha != nullptr && t0 == 0) alpha[(long long)p * H + h] = al;
                    float f = 1.f;
                    if (drop_p > 0.f) f = drop_factor(seed, (unsigned long long)p * H + h, drop_p, drop_scale);
                    s_w[h * 64 + l] = al * f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (TAB) {
                const int cnt = min(64, end - cb);
                for (int e = 0; e < cnt; ++e) gather_step_tab<VEC, NI, 1>(ft, ld_ft, s_idx, s_pos, s_t2, s_w, e, t0, nvec, hidx, acc);
            } else {
                gather_rows<VEC, NI, EU>(ft, ld_ft, s_idx, s_w, min(64, end - cb), t0, nvec, hidx, acc);
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t0 + l + 64 * i;
            if (j < nvec) {
                if (out_mode == 1) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[i][k] = leaky(acc[i][k], act_slope);
                }
                if constexpr (NX == 3) {               // the next layer's feature dropout: its GEMMs then read a plain operand
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[i][k] = ((kb[i] >> k) & 1u) ? acc[i][k] * nx.scale : 0.f;
                }
                vstore<VEC>(out + (long long)v * ld_out + (long long)j * VEC, acc[i]);
            }
        }
        if constexpr (NX == 1 || NX == 2) {
            // VEC divides 32: the VEC keep bits of a vector sit in one mask word (kb, above); the folded rows come from LDS
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = t0 + l + 64 * i;
                const int c = ((j < nvec) ? j : 0) * VEC;
                float w1[VEC], w2[VEC];
                vload<VEC>(s_wa + c, w1);
                vload<VEC>(s_wa + nx.kp + c, w2);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float xd = ((kb[i] >> k) & 1u) ? acc[i][k] : 0.f;
                    nx1 = fmaf(xd, w1[k], nx1);
                    nx2 = fmaf(xd, w2[k], nx2);
                }
            }
        }
    }
    if constexpr (NX == 1 || NX == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cc = min(F + l + 64 * i, nx.kp - 1);
            const float xd = tk[i] ? tx[i] : 0.f;
            nx1 = fmaf(xd, s_wa[cc], nx1);
            nx2 = fmaf(xd, s_wa[nx.kp + cc], nx2);
        }
        nx1 = wave_sum(nx1) * nx.scale;
        nx2 = wave_sum(nx2) * nx.scale;
        if (l == 0) { nx.a12[2 * (long long)v] = nx1; nx.a12[2 * (long long)v + 1] = nx2; }
    }
}

// NPW consecutive destination nodes per wave: the heads of their dependent-load chains (rowptr -> col -> attention terms: three round
// trips before the first feature row can be asked for) go out TOGETHER, level by level, so a wave pays the chain once per NPW nodes
// (and a workgroup its LDS staging once per 4 NPW nodes); the nodes' sweeps then run back to back.  NPW = 1: one node per wave.
template <int VEC, int NI, int NX, bool TAB = false, int NPW = 1>
__global__ __launch_bounds__(GAT_WAVES * 64, TAB ? 3 : 1) void gat_aggregate_fwd_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const int n_nodes, const float* __restrict__ ft,
    const long long ld_ft, const float* __restrict__ a_src, const float* __restrict__ a_dst, const int ld_a, const int H,
    const int D, const float slope, const float drop_p, const float drop_scale, const unsigned long long seed,
    const int out_mode, const float act_slope, float* __restrict__ out, const long long ld_out, float* __restrict__ alpha,
    const NextLogits nx, const TabSrc tab) {
    constexpr int SWH = TAB ? 4 : GAT_MAXH;            // (the table route serves H <= 4 only: its LDS goes to the T2 rows)
    __shared__ float s_w[GAT_WAVES][SWH * 64];
    __shared__ int s_idx[GAT_WAVES][64];
    __shared__ int s_pos[TAB ? GAT_WAVES : 1][64];
    __shared__ float s_stat[GAT_WAVES][2 * GAT_MAXH];

    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    extern __shared__ __attribute__((aligned(16))) float s_wa[];   // NX: the two folded rows [2][kp], shared by the workgroup's nodes;
    float* s_t2 = s_wa + ((NX == 1 || NX == 2) ? 2 * nx.kp : 0);   // TAB: behind them, the rows of T2 [vocab][ld_ft]
    if constexpr (NX == 1 || NX == 2) {
        for (int i = threadIdx.x * 4; i < 2 * nx.kp; i += GAT_WAVES * 64 * 4)
            *reinterpret_cast<float4*>(s_wa + i) = *reinterpret_cast<const float4*>(nx.wa + i);
    }
    if constexpr (TAB) {
        for (long long i = threadIdx.x * 4; i < (long long)tab.vocab * ld_ft; i += GAT_WAVES * 64 * 4)
            *reinterpret_cast<float4*>(s_t2 + i) = *reinterpret_cast<const float4*>(tab.t2 + i);
    }
    if constexpr (NPW == 1) {
        if constexpr (NX == 1 || NX == 2 || TAB) __syncthreads();     // before any wave leaves
        const int v = xcd_remap(blockIdx.x, gridDim.x) * GAT_WAVES + w;
        if (v >= n_nodes) return;
        gat_fwd_node<VEC, NI, NX, TAB>(v, l, s_w[w], s_idx[w], s_stat[w], s_wa, rowptr, col, ft, ld_ft, a_src, a_dst, ld_a, H, D, slope, drop_p,
                                       drop_scale, seed, out_mode, act_slope, out, ld_out, alpha, nx, tab, s_pos[TAB ? w : 0], s_t2);
    } else {
        static_assert(!TAB || NPW == 1, "the table route keeps one node per wave");
        // heads of the wave's NPW nodes, level by level (every level's loads are in flight together); the LDS staging above is in
        // flight beside them -- its barrier comes after
        // (wave w takes nodes w, w + 4, ... of the workgroup's 4 NPW consecutive ones: the nodes the four waves sweep at the same time are
        //  neighbours -- an egonet's siblings all gather the anchor's row -- and meet in the CU's L1 / one L2 miss; two neighbours per wave,
        //  one after the other, fetched 42 MB more)
        const int v0 = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * GAT_WAVES * NPW + w);
        NodeHead hd[NPW];
        int vk[NPW];
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            vk[k] = min(v0 + k * GAT_WAVES, n_nodes - 1);
            hd[k].beg = rowptr[vk[k]];
            hd[k].end = rowptr[vk[k] + 1];
        }
        const bool heads = H <= 4;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int deg = hd[k].end - hd[k].beg;
            const int p = hd[k].beg + ((l < deg) ? l : 0);          // clamped: the load stays unconditional when the node has an edge
            hd[k].u = (deg > 0 && deg <= 64 && heads) ? col[p] : 0;
        }
        float as[NPW][4], ad[NPW][4];
#pragma unroll
        for (int k = 0; k < NPW; ++k)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int hc = (h < H) ? h : 0;
                as[k][h] = a_src[(long long)hd[k].u * ld_a + hc];
                ad[k][h] = a_dst[(long long)vk[k] * ld_a + hc];
            }
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const bool valid = l < hd[k].end - hd[k].beg;
#pragma unroll
            for (int h = 0; h < 4; ++h) hd[k].e[h] = (valid && h < H) ? leaky(as[k][h] + ad[k][h], slope) : -INFINITY;
        }
        if constexpr (NX == 1 || NX == 2) __syncthreads();            // (every wave reaches this: none has left yet)
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            if (v0 + k * GAT_WAVES < n_nodes) {
                gat_fwd_node<VEC, NI, NX, TAB, true>(v0 + k * GAT_WAVES, l, s_w[w], s_idx[w], s_stat[w], s_wa, rowptr, col, ft, ld_ft, a_src, a_dst, ld_a, H, D, slope,
                                                     drop_p, drop_scale, seed, out_mode, act_slope, out, ld_out, alpha, nx, tab,
                                                     s_pos[TAB ? w : 0], s_t2, &hd[k]);
                __builtin_amdgcn_wave_barrier();                       // the next node re-uses this wave's LDS slots
            }
        }
    }
}

// ---- the same sweep, WALKING EGONETS (dataset.py:404-437: parents -> anchor, anchor -> siblings, self loops) ----------------------
// The kernel above fetches ft[u] once per in-edge (u -> v): an anchor's row once for itself and once per sibling, a parent's row twice
// (FETCH_SIZE 181 MB against 143 MB of rows on the training batch), and a wave asks for ONE 8-KB row per round trip.  Here a workgroup
// walks a window of npw CONSECUTIVE destination nodes, its four waves a quarter of the row each (H = 4: one head per wave), and every
// row is read once, in node order, with the next node's slice already in flight:
//   * a node whose in-list is a RUN of its predecessors and itself, [v-k .. v-1, v] (k = 0: a parent or a lone anchor; k >= 1: an anchor
//     behind its parents; k = 1 also: the first sibling behind its anchor): the predecessors' rows were accumulated into `acc` -- with
//     the coefficients of THIS node's edges, which the staging phase wrote into the predecessors' table entries -- while they were
//     walked as destinations themselves;
//   * a node whose in-list is [h, v] with h further back (the second and later siblings): the anchor's slice is kept in registers
//     (`hub`) from the moment it was walked (a window that starts behind the anchor loads it once more: the only re-read);
//   * anything else (not an egonet: other in-lists, more than 64 in-edges, none; runs that overlap) is left to gat_fwd_node, one wave
//     per node, after the walk.
// The shape is read off the destination CSR by the staging phase -- one wave per node, lane = in-edge, the softmax exactly as
// gat_fwd_node forms it -- so every edge coefficient and, per element, the order of the multiply-adds are those of the kernel above:
// `out` and `alpha` are bit-identical; the next layer's logits (nx_a12) are summed in a different order (a quarter row per wave).
// The walk itself has no branch: kinds and flags are selects, stores are unconditional (a lane past the slice repeats lane 0's vector
// and writes the same value), so the loads of node t + 1 stay in flight across the arithmetic of node t.
constexpr int EF_NODES = 32;                       // most destination nodes a workgroup walks
constexpr int EF_BACK = 64;                        // run members in front of a window's first node (in-degree <= 64)
constexpr int EF_TAB = EF_BACK + EF_NODES;
#ifndef TXE_EF_RING
#define TXE_EF_RING 4
#endif
#ifndef TXE_EF_OCC
#define TXE_EF_OCC 4
#endif
constexpr int EF_RING = TXE_EF_RING;               // row slices in flight per wave + 1
constexpr int EF_MAXE = 256;                       // in-edges of a window staged in LDS (an egonet batch has < 2 per node)
// a workgroup barrier that waits for the LDS traffic only: global loads issued before it stay in flight, global stores are not drained
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
enum { EF_RUN = 0, EF_HUBREF = 1, EF_GENERIC = 2 };

struct EgoFwdArgs {
    const int *rowptr, *col; int n_nodes;
    const float* ft; long long ld_ft; const float *a_src, *a_dst; int ld_a, D;
    float slope, drop_p, drop_scale; unsigned long long seed; int out_mode; float act_slope;
    float* out; long long ld_out; float* alpha; NextLogits nx; int npw;
    TabSrc tab;                          // TAB: ft = T, rows ft[u] = T[rid[u]] + T2[pos[u]] (a_src / a_dst: columns F.., F + H.. of the same)
};

template <int NI, int NX, bool TAB = false>
__global__ __launch_bounds__(256, (NI >= 3) ? 3 : TXE_EF_OCC) void gat_aggregate_ego_kernel(const EgoFwdArgs a) {
    __shared__ int t_kind[EF_NODES], t_hub[EF_NODES], t_ishub[EF_NODES], t_hasrun[EF_NODES];
    __shared__ float t_self[EF_NODES][4], t_hubc[EF_NODES][4];
    __shared__ int t_flag[EF_TAB], t_cnt[EF_TAB];
    __shared__ float s_nx[EF_NODES][4][2];
    __shared__ int s_nf, s_fhub, s_bad;
    __shared__ int g_idx[GAT_WAVES][64], g_pos[TAB ? GAT_WAVES : 1][64];      // gat_fwd_node's per-wave slots (nodes the walk leaves out)
    __shared__ float g_stat[GAT_WAVES][8];
    __shared__ int s_rp[EF_NODES + 1];
    __shared__ int t_rid[TAB ? EF_TAB : 1], t_posn[TAB ? EF_TAB : 1];          // TAB: table row / T2 row of the nodes [u0 - EF_BACK, u0 + EF_NODES)
    // staging tables that the generic phase no longer needs; gat_fwd_node's weight slots (4 x 256 floats) lie on top of them then
    __shared__ __attribute__((aligned(16))) float s_pool[EF_MAXE + EF_TAB * 4 + EF_NODES * 4 + EF_TAB * 4];
    static_assert(EF_MAXE + EF_TAB * 4 + EF_NODES * 4 + EF_TAB * 4 >= GAT_WAVES * 4 * 64, "gat_fwd_node's slots fit the pool");
    int* const s_col = reinterpret_cast<int*>(s_pool);                                            // [EF_MAXE]
    float (*const s_as)[4] = reinterpret_cast<float (*)[4]>(s_pool + EF_MAXE);                    // [EF_TAB]: a_src of [u0 - EF_BACK, u0 + EF_NODES)
    float (*const s_ad)[4] = reinterpret_cast<float (*)[4]>(s_pool + EF_MAXE + EF_TAB * 4);       // [EF_NODES]: a_dst of the window's nodes
    float (*const t_run)[4] = reinterpret_cast<float (*)[4]>(s_pool + EF_MAXE + EF_TAB * 4 + EF_NODES * 4);   // [EF_TAB]
    extern __shared__ __attribute__((aligned(16))) float s_wa[];   // NX 1 | 2: the next layer's two folded rows [2][kp]; TAB: T2 behind them
    float* const s_t2 = s_wa + ((NX == 1 || NX == 2) ? 2 * a.nx.kp : 0);
    constexpr int H = 4;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int u0 = b * a.npw, u1 = min(a.n_nodes, u0 + a.npw), nw = u1 - u0;     // (nw >= 1)
    const int D = a.D, F = H * D, nvec = D >> 2, c0 = w * D;
    int off[NI];
    bool live[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = l + 64 * i;
        live[i] = j < nvec;
        off[i] = c0 + 4 * (live[i] ? j : 0);
    }
    // (NX 1 | 2) the columns behind the feature part -- position embedding and zero padding, at most 128: waves 0 and 1, one per lane
    const int ct = F + w * 64 + l;
    const bool tlive = (NX == 1 || NX == 2) && w < 2 && ct < a.nx.kp;
    const int cc = tlive ? ct : 0;
    float ring[EF_RING][NI][4], txr[EF_RING];                      // position q's slice lives in slot q % EF_RING
    unsigned kbr[EF_RING][NI], tkr[EF_RING];
    // the slice of the node at window position q (q < 0: in front of the window) -- TAB: of its table row, T2's share is added at its use
    auto row_of = [&](const int q) -> const float* {
        if constexpr (TAB) return a.ft + (long long)t_rid[q + EF_BACK] * a.ld_ft;
        else return a.ft + (long long)(u0 + q) * a.ld_ft;
    };
    auto load_row = [&](const int q, float (&y)[NI][4], unsigned (&kb)[NI], float& tx, unsigned& tk) {
        const int v = u0 + q;
        const float* row = row_of(q);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            vload<4>(row + off[i], y[i]);
            kb[i] = 0xFu;
            if constexpr (NX >= 2) kb[i] = a.nx.mask[(long long)v * a.nx.mask_ld + (off[i] >> 5)] >> (off[i] & 31);
        }
        tx = 0.f; tk = 1u;
        if constexpr (NX == 1 || NX == 2) {
            tx = a.out[(long long)v * a.ld_out + cc];
            if constexpr (NX == 2) tk = (a.nx.mask[(long long)v * a.nx.mask_ld + (cc >> 5)] >> (cc & 31)) & 1u;
        }
    };
    // ---- staging, trip 1: the window's row pointers and attention terms; behind them (in flight across the whole staging phase: its
    //      barriers wait for LDS only) the first row slices of the walk ----
    if constexpr (TAB) {
        // (the table route needs one trip more before a row can be asked for: node -> table row)
        const int rp = a.rowptr[u0 + min(tid, nw)];
        const int node = min(max(u0 - EF_BACK + min(tid, EF_TAB - 1), 0), a.n_nodes - 1);
        const int rr = a.tab.rid[node], pp = a.tab.pos[node];
        for (long long i = tid * 4; i < (long long)a.tab.vocab * a.ld_ft; i += 1024)
            *reinterpret_cast<float4*>(s_t2 + i) = *reinterpret_cast<const float4*>(a.tab.t2 + i);
        if constexpr (NX == 1 || NX == 2) {
            for (int i = tid * 4; i < 2 * a.nx.kp; i += 1024) *reinterpret_cast<float4*>(s_wa + i) = *reinterpret_cast<const float4*>(a.nx.wa + i);
        }
        if (tid <= nw) s_rp[tid] = rp;
        if (tid < EF_TAB) { t_rid[tid] = rr; t_posn[tid] = pp; }
    } else {
        const int rp = a.rowptr[u0 + min(tid, nw)];
        float sa[(EF_TAB * 4 + 255) / 256];
#pragma unroll
        for (int q = 0; q < (EF_TAB * 4 + 255) / 256; ++q) {
            const int i = min(tid + 256 * q, EF_TAB * 4 - 1);
            const int node = min(max(u0 - EF_BACK + (i >> 2), 0), a.n_nodes - 1);
            sa[q] = a.a_src[(long long)node * a.ld_a + (i & 3)];
        }
        const int dn = min(tid >> 2, nw - 1) ;
        const float sd = a.a_dst[(long long)(u0 + dn) * a.ld_a + (tid & 3)];
#pragma unroll
        for (int r = 0; r < EF_RING - 1; ++r) load_row(min(r, nw - 1), ring[r], kbr[r], txr[r], tkr[r]);
        if (tid <= nw) s_rp[tid] = rp;
#pragma unroll
        for (int q = 0; q < (EF_TAB * 4 + 255) / 256; ++q) {
            const int i = tid + 256 * q;
            if (i < EF_TAB * 4) s_as[i >> 2][i & 3] = sa[q];
        }
        if (tid < EF_NODES * 4) s_ad[tid >> 2][tid & 3] = sd;
    }
    for (int i = tid; i < EF_TAB; i += 256) { t_flag[i] = 0; t_cnt[i] = 0; }
    for (int i = tid; i < EF_TAB * 4; i += 256) t_run[i >> 2][i & 3] = 0.f;
    if (tid < EF_NODES) { t_kind[tid] = EF_GENERIC; t_hub[tid] = -1; t_ishub[tid] = 0; t_hasrun[tid] = 0; }
    if (tid < EF_NODES * 4) { t_self[tid >> 2][tid & 3] = 0.f; t_hubc[tid >> 2][tid & 3] = 0.f; }
    if (tid == 0) { s_nf = 0; s_fhub = -1; s_bad = 0; }
    lds_barrier();
    // ---- trip 2: the window's in-lists (consecutive in the destination CSR) ----
    const int e0 = s_rp[0], ne = s_rp[nw] - e0;
    bool bad = ne > EF_MAXE;                                       // (not a batch of egonets: every node through gat_fwd_node)
    const int colv = (!bad && tid < ne) ? a.col[e0 + tid] : 0;
    if constexpr (TAB) {
        // attention terms from the table rows, exactly as the materialised row would hold them; behind them the first row slices
        float sa[(EF_TAB * 4 + 255) / 256];
#pragma unroll
        for (int q = 0; q < (EF_TAB * 4 + 255) / 256; ++q) {
            const int i = min(tid + 256 * q, EF_TAB * 4 - 1);
            sa[q] = a.ft[(long long)t_rid[i >> 2] * a.ld_ft + F + (i & 3)];
        }
        const int dn = EF_BACK + min(tid >> 2, nw - 1);
        const float sd = a.ft[(long long)t_rid[dn] * a.ld_ft + F + H + (tid & 3)];
#pragma unroll
        for (int r = 0; r < EF_RING - 1; ++r) load_row(min(r, nw - 1), ring[r], kbr[r], txr[r], tkr[r]);
#pragma unroll
        for (int q = 0; q < (EF_TAB * 4 + 255) / 256; ++q) {
            const int i = tid + 256 * q;
            if (i < EF_TAB * 4) s_as[i >> 2][i & 3] = sa[q] + s_t2[(long long)t_posn[i >> 2] * a.ld_ft + F + (i & 3)];
        }
        if (tid < EF_NODES * 4) s_ad[tid >> 2][tid & 3] = sd + s_t2[(long long)t_posn[dn] * a.ld_ft + F + H + (tid & 3)];
    } else if constexpr (NX == 1 || NX == 2) {                     // (the vector memory counter is in order: waiting for these also waits for
        // the row slices issued in trip 1 -- which have had a whole trip's time by now)
        for (int i = tid * 4; i < 2 * a.nx.kp; i += 1024) *reinterpret_cast<float4*>(s_wa + i) = *reinterpret_cast<const float4*>(a.nx.wa + i);
    }
    if (!bad) {
        if (tid < ne) s_col[tid] = colv;
        lds_barrier();
        // nodes with one or two in-edges (parents, siblings, lone anchors): a thread per (node, head) -- max / sum of two numbers are what
        // the wave reductions of gat_fwd_node give, bit for bit
        if (tid < nw * 4) {
            const int t = tid >> 2, h = tid & 3, v = u0 + t;
            const int beg = s_rp[t] - e0, din = s_rp[t + 1] - s_rp[t];
            if (din == 1 || din == 2) {
                const int uo = s_col[beg], us = s_col[beg + din - 1];
                int kind = EF_GENERIC;
                if (us == v) {
                    if (din == 1 || uo == v - 1) kind = EF_RUN;
                    else if (uo < v - 1 && uo >= u0 - EF_BACK) kind = EF_HUBREF;
                }
                if (kind != EF_GENERIC) {
                    const float adv = s_ad[t][h];
                    const float es = leaky(s_as[v - u0 + EF_BACK][h] + adv, a.slope);
                    const float eo = (din == 2) ? leaky(s_as[uo - u0 + EF_BACK][h] + adv, a.slope) : -INFINITY;
                    const float m = fmaxf(eo, es);
                    const float xs = __expf(es - m), xo = (din == 2) ? __expf(eo - m) : 0.f;
                    const float sm = xo + xs;
                    const int ps = e0 + beg + din - 1;
                    {
                        const float al = xs / sm;
                        if (a.alpha != nullptr) a.alpha[(long long)ps * H + h] = al;
                        float f = 1.f;
                        if (a.drop_p > 0.f) f = drop_factor(a.seed, (unsigned long long)ps * H + h, a.drop_p, a.drop_scale);
                        t_self[t][h] = al * f;
                    }
                    if (din == 2) {
                        const int po = e0 + beg;
                        const float al = xo / sm;
                        if (a.alpha != nullptr) a.alpha[(long long)po * H + h] = al;
                        float f = 1.f;
                        if (a.drop_p > 0.f) f = drop_factor(a.seed, (unsigned long long)po * H + h, a.drop_p, a.drop_scale);
                        if (kind == EF_RUN) {
                            t_run[EF_BACK + t - 1][h] = al * f;
                            if (h == 0) { t_flag[EF_BACK + t - 1] = 2; atomicAdd(&t_cnt[EF_BACK + t - 1], 1); }
                        } else t_hubc[t][h] = al * f;
                    }
                }
                if (h == 0) {
                    t_kind[t] = kind;
                    t_hub[t] = (kind == EF_HUBREF) ? uo : -1;
                    t_hasrun[t] = (kind == EF_RUN && din == 2) ? 1 : 0;
                    if (kind == EF_RUN && din == 2 && t == 0) atomicMax(&s_nf, 1);
                }
            }
        }
        // nodes with 3..64 in-edges (anchors behind their parents): a wave per node, lane = in-edge, as gat_fwd_node does it
        for (int t = w; t < nw; t += GAT_WAVES) {
            const int beg = s_rp[t] - e0, din = s_rp[t + 1] - s_rp[t];
            if (din < 3 || din > 64) continue;                                // (wave-uniform; 0 or more than 64: EF_GENERIC)
            const int v = u0 + t, first = v - (din - 1);
            const bool valid = l < din;
            const int u = s_col[beg + (valid ? l : 0)];
            if (__ballot(valid && u != first + l) != 0ull) continue;          // not a run: EF_GENERIC
            float e[4], ex[4], m[4], sm[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) e[h] = valid ? leaky(s_as[(valid ? u : v) - u0 + EF_BACK][h] + s_ad[t][h], a.slope) : -INFINITY;
#pragma unroll
            for (int h = 0; h < 4; ++h) m[h] = wave_max(e[h]);
#pragma unroll
            for (int h = 0; h < 4; ++h) ex[h] = valid ? __expf(e[h] - m[h]) : 0.f;
#pragma unroll
            for (int h = 0; h < 4; ++h) sm[h] = wave_sum(ex[h]);
            if (valid) {
                const int p = e0 + beg + l;
                const int idx = first + l - u0 + EF_BACK;                     // (run member: 1 <= idx < EF_BACK + nw)
                const bool self = l == din - 1;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const float al = ex[h] / sm[h];
                    if (a.alpha != nullptr) a.alpha[(long long)p * H + h] = al;
                    float f = 1.f;
                    if (a.drop_p > 0.f) f = drop_factor(a.seed, (unsigned long long)p * H + h, a.drop_p, a.drop_scale);
                    if (self) t_self[t][h] = al * f; else t_run[idx][h] = al * f;
                }
                if (!self) { t_flag[idx] = (l == 0) ? 2 : 1; atomicAdd(&t_cnt[idx], 1); }
            }
            if (l == 0) {
                t_kind[t] = EF_RUN; t_hasrun[t] = 1;
                if (first < u0) atomicMax(&s_nf, u0 - first);
            }
        }
        lds_barrier();
        if (tid < EF_TAB && t_cnt[tid] > 1) s_bad = 1;                        // a row wanted by two runs: one accumulator cannot serve both
        if (tid < nw && t_kind[tid] == EF_HUBREF) {
            const int hb = t_hub[tid];
            if (hb >= u0) t_ishub[hb - u0] = 1; else atomicMax(&s_fhub, hb);
        }
        lds_barrier();
        if (tid < nw && t_kind[tid] == EF_HUBREF) {                           // the slice in `hub` when this node is walked must be its hub's
            int latest = s_fhub;
            for (int j = tid - 1; j >= 0; --j) if (t_ishub[j]) { latest = u0 + j; break; }
            if (latest != t_hub[tid]) t_kind[tid] = EF_GENERIC;
        }
        lds_barrier();
        bad = s_bad != 0;
    }

    if (!bad) {
        float acc[NI][4], hub[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
        {   // the hub a window that starts behind its anchor needs
            const int fh = s_fhub;
            const int hq = (fh >= 0) ? fh - u0 : 0;
            const float* hrow = row_of(hq);
#pragma unroll
            for (int i = 0; i < NI; ++i) vload<4>(hrow + off[i], hub[i]);
            if constexpr (TAB) {
                const float* trow = s_t2 + (long long)t_posn[hq + EF_BACK] * a.ld_ft;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    float tv[4];
                    vload<4>(trow + off[i], tv);
#pragma unroll
                    for (int k = 0; k < 4; ++k) hub[i][k] += tv[k];
                }
            }
        }
        // run members in front of the window (an anchor whose parents sit in the previous window): their rows once more
        const int nf = s_nf;
        for (int f = 0; f < nf; ++f) {
            const int idx = EF_BACK - nf + f, fl = t_flag[idx];
            const float cr = t_run[idx][w];
            float y[NI][4];
            const float* frow = row_of(f - nf);
#pragma unroll
            for (int i = 0; i < NI; ++i) vload<4>(frow + off[i], y[i]);
            if constexpr (TAB) {
                const float* trow = s_t2 + (long long)t_posn[idx] * a.ld_ft;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    float tv[4];
                    vload<4>(trow + off[i], tv);
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[i][k] += tv[k];
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float tmp = fmaf(cr, y[i][k], (fl == 2) ? 0.f : acc[i][k]);
                    acc[i][k] = fl ? tmp : acc[i][k];
                }
        }
        auto step = [&](const int t, const float (&cur)[NI][4], const unsigned (&kbc)[NI], const float txc, const unsigned tkc) {
            const int v = u0 + t;
            const int kind = t_kind[t], fl = t_flag[EF_BACK + t], ish = t_ishub[t], hasrun = t_hasrun[t];
            const float cs = t_self[t][w], ch = t_hubc[t][w], cr = t_run[EF_BACK + t][w];
            float nx1 = 0.f, nx2 = 0.f;
            const float* trow = s_t2 + (TAB ? (long long)t_posn[EF_BACK + t] * a.ld_ft : 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float res[4], tv[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (TAB) vload<4>(trow + off[i], tv);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float y = TAB ? cur[i][k] + tv[k] : cur[i][k];
                    const float base = (kind == EF_HUBREF) ? fmaf(ch, hub[i][k], 0.f) : (hasrun ? acc[i][k] : 0.f);
                    float r = fmaf(cs, y, base);
                    const float tmp = fmaf(cr, y, (fl == 2) ? 0.f : acc[i][k]);
                    acc[i][k] = fl ? tmp : acc[i][k];
                    hub[i][k] = ish ? y : hub[i][k];
                    if (a.out_mode == 1) r = leaky(r, a.act_slope);
                    if constexpr (NX == 3) r = ((kbc[i] >> k) & 1u) ? r * a.nx.scale : 0.f;
                    res[k] = r;
                }
                vstore<4>(a.out + (long long)v * a.ld_out + off[i], res);
                if constexpr (NX == 1 || NX == 2) {
                    float w1[4], w2[4];
                    vload<4>(s_wa + off[i], w1);
                    vload<4>(s_wa + a.nx.kp + off[i], w2);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xd = (live[i] && ((kbc[i] >> k) & 1u)) ? res[k] : 0.f;
                        nx1 = fmaf(xd, w1[k], nx1);
                        nx2 = fmaf(xd, w2[k], nx2);
                    }
                }
            }
            if constexpr (NX == 1 || NX == 2) {
                const float xd = (tlive && tkc) ? txc : 0.f;
                nx1 = fmaf(xd, s_wa[cc], nx1);
                nx2 = fmaf(xd, s_wa[a.nx.kp + cc], nx2);
                nx1 = wave_sum(nx1);
                nx2 = wave_sum(nx2);
                if (l == 0) { s_nx[t][w][0] = nx1; s_nx[t][w][1] = nx2; }
            }
        };
        int t = 0;
        for (; t + EF_RING <= nw; t += EF_RING) {
#pragma unroll
            for (int r = 0; r < EF_RING; ++r) {
                constexpr int RN = EF_RING - 1;
                const int rn = (r + RN) % EF_RING;
                load_row(min(t + r + RN, nw - 1), ring[rn], kbr[rn], txr[rn], tkr[rn]);
                step(t + r, ring[r], kbr[r], txr[r], tkr[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < EF_RING - 1; ++r)
            if (t + r < nw) step(t + r, ring[r], kbr[r], txr[r], tkr[r]);
    }
    __syncthreads();                                                          // (the walk's stores are out: a generic node's row is rewritten below)
    if constexpr (NX == 1 || NX == 2) {
        if (!bad && tid < nw && t_kind[tid] != EF_GENERIC) {
            const float p1 = (s_nx[tid][0][0] + s_nx[tid][1][0]) + (s_nx[tid][2][0] + s_nx[tid][3][0]);
            const float p2 = (s_nx[tid][0][1] + s_nx[tid][1][1]) + (s_nx[tid][2][1] + s_nx[tid][3][1]);
            a.nx.a12[2 * (long long)(u0 + tid)] = p1 * a.nx.scale;
            a.nx.a12[2 * (long long)(u0 + tid) + 1] = p2 * a.nx.scale;
        }
    }
    for (int t = w; t < nw; t += GAT_WAVES) {
        if (bad || t_kind[t] == EF_GENERIC) {
            gat_fwd_node<4, 4, NX, TAB>(u0 + t, l, s_pool + w * 4 * 64, g_idx[w], g_stat[w], s_wa, a.rowptr, a.col, a.ft, a.ld_ft, a.a_src, a.a_dst,
                                        a.ld_a, H, a.D, a.slope, a.drop_p, a.drop_scale, a.seed, a.out_mode, a.act_slope, a.out, a.ld_out,
                                        a.alpha, a.nx, a.tab, g_pos[TAB ? w : 0], s_t2);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, destination side: d alpha (dot products), softmax + leaky-relu backward -> dz[E,H],
// d a_dst[N,H].  d_pre = gradient w.r.t. the aggregated (pre-activation) output.
// ------------------------------------------------------------------------------------------------
// Generic shape (more than 4 heads, or rows wider than one 64*NI-vector tile): one head and one edge at a time.
template <int VEC>
__global__ __launch_bounds__(GAT_WAVES * 64) void gat_bwd_edge_generic_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const int n_nodes, const float* __restrict__ ft,
    const long long ld_ft, const float* __restrict__ a_src, const float* __restrict__ a_dst, const int ld_a, const int H,
    const int D, const float slope, const float drop_p, const float drop_scale, const unsigned long long seed,
    const float* __restrict__ alpha, const float* __restrict__ d_pre, const long long ld_dpre, float* __restrict__ dz,
    float* __restrict__ d_a_dst, const int ld_da, const int n_pad) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int v = xcd_remap(blockIdx.x, gridDim.x) * GAT_WAVES + w;
    if (v >= n_nodes) return;
    for (int c = l; c < n_pad; c += 64) d_a_dst[(long long)v * ld_da + H + c] = 0.f;      // padding columns of the row
    const int beg = rowptr[v], end = rowptr[v + 1];
    const int dvec = D / VEC;
    for (int h = 0; h < H; ++h) {
        const float* drow = d_pre + (long long)v * ld_dpre + (long long)h * D;
        float sacc = 0.f;
        for (int p = beg; p < end; ++p) {
            const float* frow = ft + (long long)col[p] * ld_ft + (long long)h * D;
            float part = 0.f;
            for (int j = l; j < dvec; j += 64) {
                float a[VEC], b[VEC];
                vload<VEC>(drow + (long long)j * VEC, a);
                vload<VEC>(frow + (long long)j * VEC, b);
#pragma unroll
                for (int k = 0; k < VEC; ++k) part = fmaf(a[k], b[k], part);
            }
            const float tot = wave_sum(part);
            if (l == ((p - beg) & 63)) {
                float f = 1.f;
                if (drop_p > 0.f) f = drop_factor(seed, (unsigned long long)p * H + h, drop_p, drop_scale);
                const float dal = tot * f;
                dz[(long long)p * H + h] = dal;              // parked; the same lane re-reads it below
                sacc = fmaf(alpha[(long long)p * H + h], dal, sacc);
            }
        }
        const float S = wave_sum(sacc);
        const float ad = a_dst[(long long)v * ld_a + h];
        float dacc = 0.f;
        for (int p = beg + l; p < end; p += 64) {
            const float dal = dz[(long long)p * H + h];
            const float de = alpha[(long long)p * H + h] * (dal - S);
            const float z = a_src[(long long)col[p] * ld_a + h] + ad;
            const float g = de * (z > 0.f ? 1.f : slope);
            dz[(long long)p * H + h] = g;
            dacc += g;
        }
        dacc = wave_sum(dacc);
        if (l == 0) d_a_dst[(long long)v * ld_da + h] = dacc;
    }
}

// Shipped shapes (H <= 4, the whole H*D row in one 64*NI-vector tile).  The d_pre row of v is parked in LDS once; lane l
// owns in-edge beg + l of the current 64-edge chunk (source id, alpha, pre-activation logit fetched up front); every edge
// costs ONE sweep of NI independent 16-byte loads of ft[u], folded into per-head partial sums, reduced across the wave and
// left in lane e's register; the softmax / leaky-relu backward then runs out of registers.  Dependent chain:
// rowptr -> {col, alpha, d_pre} -> a_src -> one row sweep per edge.  Nodes with more than 64 in-edges park dz in memory.
template <int VEC, int NI>
__global__ __launch_bounds__(GAT_WAVES * 64) void gat_bwd_edge_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const int n_nodes, const float* __restrict__ ft,
    const long long ld_ft, const float* __restrict__ a_src, const float* __restrict__ a_dst, const int ld_a, const int H,
    const int D, const float slope, const float drop_p, const float drop_scale, const unsigned long long seed,
    const float* __restrict__ alpha, const float* __restrict__ d_pre, const long long ld_dpre, float* __restrict__ dz,
    float* __restrict__ d_a_dst, const int ld_da, const int n_pad) {
    __shared__ float s_d[GAT_WAVES][64 * NI * VEC];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int v = xcd_remap(blockIdx.x, gridDim.x) * GAT_WAVES + w;
    if (v >= n_nodes) return;
    for (int c = l; c < n_pad; c += 64) d_a_dst[(long long)v * ld_da + H + c] = 0.f;      // padding columns of the row
    const int beg = rowptr[v], end = rowptr[v + 1];
    if (beg == end) {                                                // no in-edge (never for egonets: self loops)
        if (l < H) d_a_dst[(long long)v * ld_da + l] = 0.f;
        return;
    }
    const int dvec = D / VEC, nvec = H * dvec;
    int off[NI];                                                     // clamped float offset of this lane's vector i
    unsigned qpack = 0;                                              // 2 bits per vector: its head; vectors past the row: weight 0
    float live[NI];
    {
        const float* drow = d_pre + (long long)v * ld_dpre;
        float t[NI][VEC];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = l + 64 * i;
            const int jc = (j < nvec) ? j : 0;
            off[i] = jc * VEC;
            qpack |= (unsigned)(jc / dvec) << (2 * i);
            live[i] = (j < nvec) ? 1.f : 0.f;
            vload<VEC>(drow + off[i], t[i]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) t[i][k] *= live[i];         // dead vectors contribute exactly 0 to every dot product
            vstore<VEC>(&s_d[w][(l + 64 * i) * VEC], t[i]);
        }
    }
    const bool one_chunk = (end - beg) <= 64;
    float sacc[4] = {0.f, 0.f, 0.f, 0.f};
    float al[4], zz[4], dal[4];
    for (int cb = beg; cb < end; cb += 64) {
        const int pl = min(cb + l, end - 1);
        const int colv = col[pl];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int h = min(t, H - 1);                             // clamped: loads stay unconditional
            al[t] = alpha[(long long)pl * H + h];
            zz[t] = a_src[(long long)colv * ld_a + h] + a_dst[(long long)v * ld_a + h];
            dal[t] = 0.f;
        }
        const int cnt = min(64, end - cb);
        for (int e = 0; e < cnt; ++e) {
            const float* frow = ft + (long long)__builtin_amdgcn_readlane(colv, e) * ld_ft;
            float b[NI][VEC];
#pragma unroll
            for (int i = 0; i < NI; ++i) vload<VEC>(frow + off[i], b[i]);
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float a[VEC];
                vload<VEC>(&s_d[w][(l + 64 * i) * VEC], a);
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < VEC; ++k) d = fmaf(a[k], b[i][k], d);
                const unsigned qi = (qpack >> (2 * i)) & 3u;
#pragma unroll
                for (int t = 0; t < 4; ++t) part[t] += (qi == (unsigned)t) ? d : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < H) {                                          // wave-uniform
                    const float tot = wave_sum(part[t]);
                    if (l == e) {
                        float f = 1.f;
                        if (drop_p > 0.f) f = drop_factor(seed, (unsigned long long)(cb + e) * H + t, drop_p, drop_scale);
                        dal[t] = tot * f;
                    }
                }
            }
        }
        const bool lvalid = cb + l < end;
        if (one_chunk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < H) {
                    const float S = wave_sum(lvalid ? al[t] * dal[t] : 0.f);
                    const float g = lvalid ? al[t] * (dal[t] - S) * (zz[t] > 0.f ? 1.f : slope) : 0.f;
                    if (lvalid) dz[(long long)(cb + l) * H + t] = g;
                    const float dacc = wave_sum(g);
                    if (l == 0) d_a_dst[(long long)v * ld_da + t] = dacc;
                }
            }
        } else if (lvalid) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < H) {
                    dz[(long long)(cb + l) * H + t] = dal[t];          // parked; the same lane re-reads it below
                    sacc[t] = fmaf(al[t], dal[t], sacc[t]);
                }
            }
        }
    }
    if (!one_chunk) {
        for (int t = 0; t < H; ++t) {
            const float S = wave_sum(sacc[t]);
            const float ad = a_dst[(long long)v * ld_a + t];
            float dacc = 0.f;
            for (int p = beg + l; p < end; p += 64) {
                const float de = alpha[(long long)p * H + t] * (dz[(long long)p * H + t] - S);
                const float z = a_src[(long long)col[p] * ld_a + t] + ad;
                const float g = de * (z > 0.f ? 1.f : slope);
                dz[(long long)p * H + t] = g;
                dacc += g;
            }
            dacc = wave_sum(dacc);
            if (l == 0) d_a_dst[(long long)v * ld_da + t] = dacc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, source side (source-sorted CSR, atomic free):
//   d_ft[u] = sum_{e=(u->v)} a_drop[e] * d_pre[v]        d_a_src[u,h] = sum_{e=(u->v)} dz[e,h]
// pos_out[j] = position of out-edge j in the destination-sorted order (where alpha / dz live).
// ------------------------------------------------------------------------------------------------
// one wave, one source node u, the whole H*D row
template <int VEC, int NI>
__device__ __forceinline__ void bwd_node_wave(const int u, const int l, float* __restrict__ s_w, int* __restrict__ s_idx,
    const int* __restrict__ rowptr_out, const int* __restrict__ col_dst, const int* __restrict__ pos_out,
    const float* __restrict__ alpha, const float* __restrict__ dz, const int H, const int D, const float drop_p,
    const float drop_scale, const unsigned long long seed, const float* __restrict__ d_pre, const long long ld_dpre,
    float* __restrict__ d_ft, const long long ld_dft, float* __restrict__ d_a_src, const int ld_da) {
    const int beg = rowptr_out[u], end = rowptr_out[u + 1];

    for (int h = 0; h < H; ++h) {
        float acc = 0.f;
        for (int j = beg + l; j < end; j += 64) acc += dz[(long long)pos_out[j] * H + h];
        acc = wave_sum(acc);
        if (l == 0) d_a_src[(long long)u * ld_da + h] = acc;
    }

    const int F = H * D, nvec = F / VEC;
    for (int t0 = 0; t0 < nvec; t0 += 64 * NI) {
        int hidx[NI];
        float acc[NI][VEC];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t0 + l + 64 * i;
            hidx[i] = (j < nvec) ? (j * VEC) / D : 0;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
        }
        for (int cb = beg; cb < end; cb += 64) {
            const int j = cb + l;
            if (j < end) {
                const int q = pos_out[j];
                s_idx[l] = col_dst[j];
                for (int h = 0; h < H; ++h) {
                    float f = 1.f;
                    if (drop_p > 0.f) f = drop_factor(seed, (unsigned long long)q * H + h, drop_p, drop_scale);
                    s_w[h * 64 + l] = alpha[(long long)q * H + h] * f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            gather_rows<VEC, NI, (NI >= 8 ? 1 : 2)>(d_pre, ld_dpre, s_idx, s_w, min(64, end - cb), t0, nvec, hidx, acc);
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t0 + l + 64 * i;
            if (j < nvec) vstore<VEC>(d_ft + (long long)u * ld_dft + (long long)j * VEC, acc[i]);
        }
    }
}


template <int VEC, int NI>
__global__ __launch_bounds__(GAT_WAVES * 64) void gat_bwd_node_kernel(
    const int* __restrict__ rowptr_out, const int* __restrict__ col_dst, const int* __restrict__ pos_out, const int n_nodes,
    const float* __restrict__ alpha, const float* __restrict__ dz, const int H, const int D, const float drop_p,
    const float drop_scale, const unsigned long long seed, const float* __restrict__ d_pre, const long long ld_dpre,
    float* __restrict__ d_ft, const long long ld_dft, float* __restrict__ d_a_src, const int ld_da) {
    __shared__ float s_w[GAT_WAVES][GAT_MAXH * 64];
    __shared__ int s_idx[GAT_WAVES][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int u = xcd_remap(blockIdx.x, gridDim.x) * GAT_WAVES + w;
    if (u >= n_nodes) return;
    bwd_node_wave<VEC, NI>(u, l, s_w[w], s_idx[w], rowptr_out, col_dst, pos_out, alpha, dz, H, D, drop_p, drop_scale, seed, d_pre, ld_dpre,
                           d_ft, ld_dft, d_a_src, ld_da);
}

// Workgroup-cooperative variant for wide rows: the 4 waves of a workgroup share each of its 4 source nodes, every wave
// owning a quarter of the H*D row.  A node with many out-edges (the anchor of an egonet feeds up to 50 siblings) then
// costs each wave a quarter of the row per edge instead of serialising one wave on the whole row (measured 2.4x on the
// MAG layer-0 backward).  Requires nvec = H*D/VEC <= 4*64*SLICE_NI.
template <int VEC>
__global__ __launch_bounds__(GAT_WAVES * 64) void gat_bwd_node_split_kernel(
    const int* __restrict__ rowptr_out, const int* __restrict__ col_dst, const int* __restrict__ pos_out, const int n_nodes,
    const float* __restrict__ alpha, const float* __restrict__ dz, const int H, const int D, const float drop_p,
    const float drop_scale, const unsigned long long seed, const float* __restrict__ d_pre, const long long ld_dpre,
    float* __restrict__ d_ft, const long long ld_dft, float* __restrict__ d_a_src, const int ld_da) {
    __shared__ float s_w[GAT_WAVES][GAT_MAXH * 64];
    __shared__ int s_idx[GAT_WAVES][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int node0 = xcd_remap(blockIdx.x, gridDim.x) * GAT_WAVES;
    const int F = H * D, nvec = F / VEC;
    const int q = (nvec + GAT_WAVES - 1) / GAT_WAVES;
    const int j0 = w * q, j1 = min(nvec, j0 + q);
    int hidx[SLICE_NI];
#pragma unroll
    for (int i = 0; i < SLICE_NI; ++i) {
        const int j = j0 + l + 64 * i;
        hidx[i] = (j < j1) ? (j * VEC) / D : 0;
    }
    // The out-edge lists of all 4 nodes are fetched up front (first 64 edges each, H <= 4): rowptr -> {pos_out, col_dst} ->
    // {alpha, dz} is paid once per workgroup instead of once per node, and the 4 gathers then run back to back.
    const int n_edges = rowptr_out[n_nodes];
    const bool pre = (H <= 4) && (n_edges > 0);
    int nbeg[GAT_WAVES], nend[GAT_WAVES], pidx[GAT_WAVES], pq[GAT_WAVES];
    float pw[GAT_WAVES][4];
#pragma unroll
    for (int t = 0; t < GAT_WAVES; ++t) {
        const int u = min(node0 + t, n_nodes - 1);
        nbeg[t] = rowptr_out[u];
        nend[t] = (node0 + t < n_nodes) ? rowptr_out[u + 1] : nbeg[t];
    }
    // Light workgroup (the usual one: an egonet's parents and siblings have 1-2 out-edges, its anchor a handful): one wave per
    // node sweeps its whole row with 8 independent 16-byte loads per edge -- the cooperative path below has two loads in flight
    // per lane and edge and walks its four nodes one after the other, which only pays when a node feeds many destinations.
    if (SPLIT_HYBRID) {
        int maxdeg = 0;
#pragma unroll
        for (int t = 0; t < GAT_WAVES; ++t) maxdeg = max(maxdeg, nend[t] - nbeg[t]);
        if (maxdeg <= SPLIT_LIGHT_DEG && nvec <= 512) {                  // workgroup-uniform
            if (node0 + w < n_nodes)
                bwd_node_wave<VEC, 8>(node0 + w, l, s_w[w], s_idx[w], rowptr_out, col_dst, pos_out, alpha, dz, H, D, drop_p, drop_scale, seed,
                                      d_pre, ld_dpre, d_ft, ld_dft, d_a_src, ld_da);
            return;
        }
    }
    if (pre) {
#pragma unroll
        for (int t = 0; t < GAT_WAVES; ++t) {
            const int j = min(nbeg[t] + l, n_edges - 1);                // clamped: loads stay unconditional
            pq[t] = pos_out[j];
            pidx[t] = col_dst[j];
        }
#pragma unroll
        for (int t = 0; t < GAT_WAVES; ++t)
#pragma unroll
            for (int h = 0; h < 4; ++h) pw[t][h] = alpha[(long long)pq[t] * H + min(h, H - 1)];
        // d_a_src of node node0 + w: this wave's own reduction, from the same prefetch
        int qw = pq[0], bw = nbeg[0], ew = nend[0];
#pragma unroll
        for (int t = 1; t < GAT_WAVES; ++t) {
            qw = (w == t) ? pq[t] : qw;
            bw = (w == t) ? nbeg[t] : bw;
            ew = (w == t) ? nend[t] : ew;
        }
        if (node0 + w < n_nodes) {
            for (int h = 0; h < H; ++h) {
                float a = (bw + l < ew) ? dz[(long long)qw * H + h] : 0.f;
                for (int j = bw + 64 + l; j < ew; j += 64) a += dz[(long long)pos_out[j] * H + h];
                a = wave_sum(a);
                if (l == 0) d_a_src[(long long)(node0 + w) * ld_da + h] = a;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < GAT_WAVES; ++t) {
        const int u = node0 + t;
        if (u >= n_nodes) break;                                   // workgroup-uniform
        const int beg = nbeg[t], end = nend[t];
        if (!pre && w == t) {                                      // one wave per node does the tiny d_a_src reduction
            for (int h = 0; h < H; ++h) {
                float a = 0.f;
                for (int j = beg + l; j < end; j += 64) a += dz[(long long)pos_out[j] * H + h];
                a = wave_sum(a);
                if (l == 0) d_a_src[(long long)u * ld_da + h] = a;
            }
        }
        float acc[SLICE_NI][VEC];
#pragma unroll
        for (int i = 0; i < SLICE_NI; ++i)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
        for (int cb = beg; cb < end; cb += 64) {
            const int j = cb + l;
            if (j < end) {
                if (pre && cb == beg) {
                    s_idx[w][l] = pidx[t];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        if (h < H) {
                            float f = 1.f;
                            if (drop_p > 0.f) f = drop_factor(seed, (unsigned long long)pq[t] * H + h, drop_p, drop_scale);
                            s_w[w][h * 64 + l] = pw[t][h] * f;
                        }
                    }
                } else {
                    const int qd = pos_out[j];
                    s_idx[w][l] = col_dst[j];
                    for (int h = 0; h < H; ++h) {
                        float f = 1.f;
                        if (drop_p > 0.f) f = drop_factor(seed, (unsigned long long)qd * H + h, drop_p, drop_scale);
                        s_w[w][h * 64 + l] = alpha[(long long)qd * H + h] * f;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            gather_rows<VEC, SLICE_NI, 4>(d_pre, ld_dpre, s_idx[w], s_w[w], min(64, end - cb), j0, j1, hidx, acc);
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int i = 0; i < SLICE_NI; ++i) {
            const int j = j0 + l + 64 * i;
            if (j < j1) vstore<VEC>(d_ft + (long long)u * ld_dft + (long long)j * VEC, acc[i]);
        }
    }
}

// d_pre = d_out * leaky'(out_act)   (out_act = leaky(out): same sign as out)
__global__ void leaky_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ out_act, float slope,
                                 long long n, float* __restrict__ d_pre) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        d_pre[i] = d_out[i] * (out_act[i] > 0.f ? 1.f : slope);
}

// mean over heads: y[n,d] = (1/H) sum_h x[n,h,d]  (model_zoo.py:219 `.mean(1)`), and its backward.
__global__ void head_mean_kernel(const float* __restrict__ x, int H, int D, long long n_rows, float* __restrict__ y) {
    const long long total = n_rows * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / D;
        const int d = (int)(i % D);
        float s = 0.f;
        for (int h = 0; h < H; ++h) s += x[(r * H + h) * D + d];
        y[i] = s / (float)H;
    }
}
__global__ void head_mean_bwd_kernel(const float* __restrict__ dy, int H, int D, long long n_rows, float* __restrict__ dx) {
    const long long total = n_rows * H * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ((long long)H * D);
        const int d = (int)(i % D);
        dx[i] = dy[r * D + d] / (float)H;
    }
}

// name as rocprofv3 prints it, e.g. "gat_aggregate_fwd_kernel<4, 8>"
struct KName {
    char s[64];
    KName(const char* base, int a, int b) { snprintf(s, sizeof(s), "%s<%d, %d>", base, a, b); }
    KName(const char* base, int a) { snprintf(s, sizeof(s), "%s<%d>", base, a); }
    KName(const char* base, int a, int b, int c) { snprintf(s, sizeof(s), "%s<%d, %d, %d>", base, a, b, c); }
    KName(const char* base, int a, int b, bool c) { snprintf(s, sizeof(s), "%s<%d, %d, %s>", base, a, b, c ? "true" : "false"); }
    KName(const char* base, int a, int b, int c, bool d) { snprintf(s, sizeof(s), "%s<%d, %d, %d, %s>", base, a, b, c, d ? "true" : "false"); }
    KName(const char* base, int a, int b, int c, bool d, int e) { snprintf(s, sizeof(s), "%s<%d, %d, %d, %s, %d>", base, a, b, c, d ? "true" : "false", e); }
};

#define TXE_DISPATCH_VEC_NI(vec, ni, LAUNCH)                                     \
    do {                                                                         \
        if (vec == 4) { if (ni == 8) LAUNCH(4, 8); else if (ni == 4) LAUNCH(4, 4); else LAUNCH(4, 2); } \
        else if (vec == 2) { if (ni == 8) LAUNCH(2, 8); else if (ni == 4) LAUNCH(2, 4); else LAUNCH(2, 2); } \
        else { if (ni == 8) LAUNCH(1, 8); else if (ni == 4) LAUNCH(1, 4); else LAUNCH(1, 2); } \
    } while (0)

// destination nodes per wave of the forward sweep (gat_aggregate_fwd_kernel's NPW): two -- 97 -> 93 us in the training step, 77 -> 72 us
// stand-alone (tools/agg_fwd_variants.py); the entry point's `npw` argument (1 | 2) overrides -- the two are bit-equal, a parity test
static inline int fwd_nodes_per_wave(int n_nodes, int forced) {
    if (forced == 1 || forced == 2) return forced;
    return n_nodes >= 4096 ? 2 : 1;       // (4 per wave was measured too: 173 VGPRs = two waves per SIMD, 119 us against 93 us)
}

// destination nodes per workgroup of the egonet walk: one round of workgroups where the batch allows it (four per CU), a workgroup
// costing about (nodes + 4)
int device_cu_count();     // txe_profile.hip
static inline int ef_nodes_per_wg(int n_nodes, int occupancy) {
    const int slots = occupancy * device_cu_count();
    int best = 16;
    double best_cost = 1e30;
    for (int npw = 12; npw <= EF_NODES; npw += 2) {
        const long long blocks = ((long long)n_nodes + npw - 1) / npw;
        const double cost = (double)((blocks + slots - 1) / slots) * (npw + 4.0);
        if (cost < best_cost) { best_cost = cost; best = npw; }
    }
    return best;
}

static inline int pick_vec(int D, long long ld1, long long ld2, const void* p1, const void* p2) {
    auto al = [](const void* p, int bytes) { return ((uintptr_t)p % bytes) == 0; };
    if (D % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0 && al(p1, 16) && al(p2, 16)) return 4;
    if (D % 2 == 0 && ld1 % 2 == 0 && ld2 % 2 == 0 && al(p1, 8) && al(p2, 8)) return 2;
    return 1;
}

}  // namespace txe

using namespace txe;

extern "C" {

int txe_gat_aggregate_fwd(const int* rowptr_in, const int* col_src, int n_nodes, const float* ft, long long ld_ft,
                          const float* a_src, const float* a_dst, int ld_a, int H, int D, float attn_slope, float attn_drop_p,
                          unsigned long long seed, int out_mode, float act_slope, float* out, long long ld_out, float* alpha,
                          const float* nx_wa, int nx_kp, const unsigned* nx_mask, float nx_feat_drop_p, float* nx_a12, int npw_req, void* stream) {
    if (n_nodes < 0 || H < 1 || H > GAT_MAXH || D < 1 || !rowptr_in || !ft || !a_src || !a_dst || !out || npw_req < 0 ||
        (npw_req > 4 && npw_req < 8) || npw_req > EF_NODES)
        return TXE_ERR_ARG;
    if (out_mode != 0 && out_mode != 1) return TXE_ERR_ARG;
    if (attn_drop_p < 0.f || attn_drop_p >= 1.f) return TXE_ERR_ARG;
    if (nx_a12 && (!nx_wa || nx_kp < H * D || nx_kp - H * D > 128 || (nx_kp & 31) || ld_out != nx_kp || nx_feat_drop_p < 0.f || nx_feat_drop_p >= 1.f ||
                   (nx_feat_drop_p > 0.f && !nx_mask)))
        return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    const float scale = 1.f / (1.f - attn_drop_p);
    hipStream_t s = (hipStream_t)stream;
    const int vec = pick_vec(D, ld_ft, ld_out, ft, out);
    if (nx_a12 && (vec != 4 || ((uintptr_t)nx_wa & 15) || nx_kp > 4096)) return TXE_ERR_ARG;   // 16-byte layout; rows fit 32 KB of LDS
    // nx_mask WITHOUT nx_a12: only the next layer's feature dropout is applied to the rows written (that layer's GEMMs then read a plain
    // operand; its preparation drops the position columns the same way): 16-byte rows, nx_kp = ld_out = that layer's padded width
    const bool out_drop = !nx_a12 && nx_mask && nx_feat_drop_p > 0.f;
    if (out_drop && (vec != 4 || nx_kp < H * D || (nx_kp & 31) || ld_out != nx_kp || nx_feat_drop_p >= 1.f)) return TXE_ERR_ARG;
    NextLogits nx;
    nx.wa = nx_wa; nx.mask = (nx_feat_drop_p > 0.f) ? nx_mask : nullptr; nx.a12 = nx_a12;
    nx.scale = 1.f / (1.f - ((nx_a12 || out_drop) ? nx_feat_drop_p : 0.f)); nx.kp = nx_kp; nx.mask_ld = nx_kp / 32;
    // algorithmic (compulsory) bytes, SURVEY 8d: read ft + write out + a_src/a_dst + CSR (+ alpha when kept for backward);
    // the edge count is not known here (device rowptr), the E-proportional terms are added by the caller-side model
    // (16-byte rows of 257-320 or 513-640 vectors -- SemEval's 4 x 600 columns are 600 -- take 5 per lane: two passes over 320 lane slots
    //  instead of two over 512 whose clamped loads still issue)
    const int nvec = H * D / vec;
    // the egonet walk (gat_aggregate_ego_kernel): four heads, 16-byte rows of at most 192 vectors per head; npw_req 0 = when the batch
    // fills the chip, 3 | 8..32 = forced (with that many nodes per workgroup), 1 | 2 | 4 = never
    const bool ego_fits = H == 4 && vec == 4 && (D & 3) == 0 && D / 4 <= 192 && (!nx_wa || nx_kp <= 4096);
    if ((npw_req == 3 || npw_req >= 8) && !ego_fits) return TXE_ERR_ARG;
    if (ego_fits && (npw_req == 3 || npw_req >= 8 || (npw_req == 0 && n_nodes >= 4096))) {
        EgoFwdArgs ea;
        ea.rowptr = rowptr_in; ea.col = col_src; ea.n_nodes = n_nodes; ea.ft = ft; ea.ld_ft = ld_ft; ea.a_src = a_src; ea.a_dst = a_dst;
        ea.ld_a = ld_a; ea.D = D; ea.slope = attn_slope; ea.drop_p = attn_drop_p; ea.drop_scale = scale; ea.seed = seed;
        ea.out_mode = out_mode; ea.act_slope = act_slope; ea.out = out; ea.ld_out = ld_out; ea.alpha = alpha; ea.nx = nx;
        ea.tab = TabSrc{nullptr, nullptr, nullptr, 0};
        ea.npw = (npw_req >= 8) ? npw_req : ef_nodes_per_wg(n_nodes, (D / 4 <= 128) ? TXE_EF_OCC : 3);
        const int nbe = (n_nodes + ea.npw - 1) / ea.npw;
        const int nie = (D / 4 <= 128) ? 2 : 3;
        const int mode = nx_a12 ? (nx.mask ? 2 : 1) : (out_drop ? 3 : 0);
        const size_t lds = nx_a12 ? (size_t)2 * nx_kp * sizeof(float) : 0;
        const KName kn("gat_aggregate_ego_kernel", nie, mode);
        ProfScope prof(kn.s, s, 4.0 * (2.0 * n_nodes * (double)H * D + 2.0 * n_nodes * H + n_nodes + 1), 1);
#define TXE_LE(I, M) hipLaunchKernelGGL((gat_aggregate_ego_kernel<I, M>), dim3(nbe), dim3(256), lds, s, ea)
#define TXE_LEM(I) do { if (mode == 0) TXE_LE(I, 0); else if (mode == 1) TXE_LE(I, 1); else if (mode == 2) TXE_LE(I, 2); else TXE_LE(I, 3); } while (0)
        if (nie == 2) TXE_LEM(2); else TXE_LEM(3);
#undef TXE_LEM
#undef TXE_LE
        TXE_CHECK_LAUNCH();
        return TXE_OK;
    }
    const int ni = (vec == 4 && ((nvec > 256 && nvec <= 320) || (nvec > 512 && nvec <= 640))) ? 5 : pick_ni(nvec);
    const int npw = fwd_nodes_per_wave(n_nodes, npw_req);
    const int nb = (n_nodes + GAT_WAVES * npw - 1) / (GAT_WAVES * npw);
    const KName kn("gat_aggregate_fwd_kernel", vec, ni, nx_a12 ? (nx.mask ? 2 : 1) : (out_drop ? 3 : 0), false, npw);
    ProfScope prof(kn.s, s, 4.0 * (2.0 * n_nodes * (double)H * D + 2.0 * n_nodes * H + n_nodes + 1), 1);
#define TXE_LK(V, I, M, P, LDS)                                                                                                   \
    hipLaunchKernelGGL((gat_aggregate_fwd_kernel<V, I, M, false, P>), dim3(nb), dim3(GAT_WAVES * 64), LDS, s, rowptr_in, col_src, n_nodes, \
                       ft, ld_ft, a_src, a_dst, ld_a, H, D, attn_slope, attn_drop_p, scale, seed, out_mode, act_slope,            \
                       out, ld_out, alpha, nx, TabSrc{nullptr, nullptr, nullptr, 0})
#define TXE_LP(V, I, M, LDS) do { if (npw == 2) TXE_LK(V, I, M, 2, LDS); else TXE_LK(V, I, M, 1, LDS); } while (0)
#define TXE_L(V, I) TXE_LP(V, I, 0, 0)
#define TXE_LXM(I, M) TXE_LP(4, I, M, (size_t)2 * nx_kp * sizeof(float))
#define TXE_LX(I) do { if (nx.mask) TXE_LXM(I, 2); else TXE_LXM(I, 1); } while (0)
#define TXE_LD(I) TXE_LP(4, I, 3, 0)
    if (nx_a12) { if (ni == 8) TXE_LX(8); else if (ni == 5) TXE_LX(5); else if (ni == 4) TXE_LX(4); else TXE_LX(2); }
    else if (out_drop) { if (ni == 8) TXE_LD(8); else if (ni == 5) TXE_LD(5); else if (ni == 4) TXE_LD(4); else TXE_LD(2); }
    else if (ni == 5) TXE_L(4, 5);
    else TXE_DISPATCH_VEC_NI(vec, ni, TXE_L);
#undef TXE_LD
#undef TXE_LX
#undef TXE_LXM
#undef TXE_L
#undef TXE_LP
#undef TXE_LK
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// Eval-mode first layer on rows of a projected feature table: ft[u] = T[rid[u]] + T2[pos[u]] is formed inside the
// sweep instead of being written out and read back (MAG-Full: 1.1 M rows of 8 KB).  No dropout, no alpha kept: inference only.
static size_t table_lds_bytes(long long ld_t, int vocab, int nx_kp) { return ((size_t)vocab * ld_t + 2 * (size_t)nx_kp) * sizeof(float); }

int txe_gat_aggregate_table_supported(int H, int D, long long ld_t, int vocab, int nx_kp) {
    return (H >= 1 && H <= 4 && D >= 4 && (D & 3) == 0 && (ld_t & 3) == 0 && ld_t >= (long long)H * D + 2 * H && vocab >= 1 && nx_kp >= 0 &&
            (nx_kp & 3) == 0 && table_lds_bytes(ld_t, vocab, nx_kp) <= 56 * 1024) ? 1 : 0;
}

int txe_gat_aggregate_table_fwd(const int* rowptr_in, const int* col_src, int n_nodes, const float* T, long long ld_t, const int* rid,
                                const float* T2, const int* pos, int vocab, int H, int D, float attn_slope, int out_mode,
                                float act_slope, float* out, long long ld_out, const float* nx_wa, int nx_kp, float* nx_a12,
                                int npw_req, void* stream) {
    if (n_nodes < 0 || !rowptr_in || !T || !rid || !T2 || !pos || !out || (out_mode != 0 && out_mode != 1)) return TXE_ERR_ARG;
    if (npw_req < 0 || (npw_req > 4 && npw_req < 8) || npw_req > EF_NODES || npw_req == 2) return TXE_ERR_ARG;
    if (!txe_gat_aggregate_table_supported(H, D, ld_t, vocab, nx_a12 ? nx_kp : 0)) return TXE_ERR_ARG;
    if ((ld_out & 3) || (((uintptr_t)T | (uintptr_t)T2 | (uintptr_t)out) & 15)) return TXE_ERR_ARG;
    if (nx_a12 && (!nx_wa || nx_kp < H * D || nx_kp - H * D > 128 || (nx_kp & 31) || ld_out != nx_kp || ((uintptr_t)nx_wa & 15))) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (n_nodes + GAT_WAVES - 1) / GAT_WAVES;
    NextLogits nx;
    nx.wa = nx_wa; nx.mask = nullptr; nx.a12 = nx_a12; nx.scale = 1.f; nx.kp = nx_a12 ? nx_kp : 0; nx.mask_ld = nx.kp / 32;
    TabSrc tab;
    tab.rid = rid; tab.pos = pos; tab.t2 = T2; tab.vocab = vocab;
    const size_t lds = table_lds_bytes(ld_t, vocab, nx.kp);
    // the egonet walk (gat_aggregate_ego_kernel<.., TAB>), as in txe_gat_aggregate_fwd: npw_req 0 = for batches that fill the chip,
    // 3 | 8..32 = forced, 1 | 4 = one wave per node
    const bool ego_fits = H == 4 && D / 4 <= 192 && lds + 12 * 1024 <= 64 * 1024;
    if ((npw_req == 3 || npw_req >= 8) && !ego_fits) return TXE_ERR_ARG;
    if (ego_fits && (npw_req == 3 || npw_req >= 8 || (npw_req == 0 && n_nodes >= 4096))) {
        EgoFwdArgs ea;
        ea.rowptr = rowptr_in; ea.col = col_src; ea.n_nodes = n_nodes; ea.ft = T; ea.ld_ft = ld_t; ea.a_src = T; ea.a_dst = T;
        ea.ld_a = 0; ea.D = D; ea.slope = attn_slope; ea.drop_p = 0.f; ea.drop_scale = 1.f; ea.seed = 0ull;
        ea.out_mode = out_mode; ea.act_slope = act_slope; ea.out = out; ea.ld_out = ld_out; ea.alpha = nullptr; ea.nx = nx; ea.tab = tab;
        const int nie = (D / 4 <= 128) ? 2 : 3;
        const int occ_lds = (int)((160 * 1024) / (lds + 12 * 1024)), occ = ((nie == 2) ? TXE_EF_OCC : 3) < occ_lds ? ((nie == 2) ? TXE_EF_OCC : 3) : occ_lds;
        ea.npw = (npw_req >= 8) ? npw_req : ef_nodes_per_wg(n_nodes, occ);
        const int nbe = (n_nodes + ea.npw - 1) / ea.npw;
        const KName kn("gat_aggregate_ego_kernel", nie, nx_a12 ? 1 : 0, true);
        ProfScope prof(kn.s, s, 4.0 * (2.0 * n_nodes * (double)H * D + 2.0 * n_nodes * H + 3.0 * n_nodes + 1), 1);
#define TXE_LET(I, X) hipLaunchKernelGGL((gat_aggregate_ego_kernel<I, X, true>), dim3(nbe), dim3(256), lds, s, ea)
        if (nx_a12) { if (nie == 2) TXE_LET(2, 1); else TXE_LET(3, 1); }
        else { if (nie == 2) TXE_LET(2, 0); else TXE_LET(3, 0); }
#undef TXE_LET
        TXE_CHECK_LAUNCH();
        return TXE_OK;
    }
    const int ni = pick_ni(H * D / 4);
    const KName kn("gat_aggregate_fwd_kernel", 4, ni, nx_a12 ? 1 : 0, true, 1);
    ProfScope prof(kn.s, s, 4.0 * (2.0 * n_nodes * (double)H * D + 2.0 * n_nodes * H + 3.0 * n_nodes + 1), 1);
#define TXE_LT(I, X)                                                                                                              \
    hipLaunchKernelGGL((gat_aggregate_fwd_kernel<4, I, X, true>), dim3(nb), dim3(GAT_WAVES * 64), lds, s, rowptr_in, col_src, n_nodes, T, \
                       ld_t, T, T, 0, H, D, attn_slope, 0.f, 1.f, 0ull, out_mode, act_slope, out, ld_out, (float*)nullptr, nx, tab)
    if (nx_a12) { if (ni == 8) TXE_LT(8, 1); else if (ni == 4) TXE_LT(4, 1); else TXE_LT(2, 1); }
    else { if (ni == 8) TXE_LT(8, 0); else if (ni == 4) TXE_LT(4, 0); else TXE_LT(2, 0); }
#undef TXE_LT
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_gat_aggregate_bwd(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst,
                          const int* pos_out, int n_nodes, const float* ft, long long ld_ft, const float* a_src,
                          const float* a_dst, int ld_a, int H, int D, float attn_slope, float attn_drop_p,
                          unsigned long long seed, const float* alpha, const float* d_pre, long long ld_dpre, float* d_ft,
                          long long ld_dft, float* d_a_src, float* d_a_dst, int ld_da, float* dz_ws, int n_pad, void* stream) {
    if (n_nodes < 0 || H < 1 || H > GAT_MAXH || D < 1 || n_pad < 0) return TXE_ERR_ARG;
    if (!rowptr_in || !rowptr_out || !ft || !alpha || !d_pre || !d_ft || !d_a_src || !d_a_dst || !dz_ws) return TXE_ERR_ARG;
    if (attn_drop_p < 0.f || attn_drop_p >= 1.f) return TXE_ERR_ARG;
    if (n_nodes == 0) return TXE_OK;
    const float scale = 1.f / (1.f - attn_drop_p);
    const int nb = (n_nodes + GAT_WAVES - 1) / GAT_WAVES;
    hipStream_t s = (hipStream_t)stream;
    const int v1 = pick_vec(D, ld_ft, ld_dpre, ft, d_pre);
    const int nvec1 = H * D / v1;
    if (H <= 4 && nvec1 <= 512) {
    const int ni1 = pick_ni(nvec1);
    const KName kn("gat_bwd_edge_kernel", v1, ni1);
    ProfScope prof(kn.s, s, 4.0 * (2.0 * n_nodes * (double)H * D + 3.0 * n_nodes * H), 1);   // read ft + d_pre
#define TXE_L(V, I)                                                                                                        \
    hipLaunchKernelGGL((gat_bwd_edge_kernel<V, I>), dim3(nb), dim3(GAT_WAVES * 64), 0, s, rowptr_in, col_src, n_nodes, ft, \
                       ld_ft, a_src, a_dst, ld_a, H, D, attn_slope, attn_drop_p, scale, seed, alpha, d_pre, ld_dpre,       \
                       dz_ws, d_a_dst, ld_da, n_pad)
    TXE_DISPATCH_VEC_NI(v1, ni1, TXE_L);
#undef TXE_L
    } else {
    const KName kn("gat_bwd_edge_generic_kernel", v1);
    ProfScope prof(kn.s, s, 4.0 * (2.0 * n_nodes * (double)H * D + 3.0 * n_nodes * H), 1);
#define TXE_L(V)                                                                                                           \
    hipLaunchKernelGGL((gat_bwd_edge_generic_kernel<V>), dim3(nb), dim3(GAT_WAVES * 64), 0, s, rowptr_in, col_src, n_nodes, \
                       ft, ld_ft, a_src, a_dst, ld_a, H, D, attn_slope, attn_drop_p, scale, seed, alpha, d_pre, ld_dpre,    \
                       dz_ws, d_a_dst, ld_da, n_pad)
    if (v1 == 4) TXE_L(4); else if (v1 == 2) TXE_L(2); else TXE_L(1);
#undef TXE_L
    }
    TXE_CHECK_LAUNCH();
    const int v2 = pick_vec(D, ld_dpre, ld_dft, d_pre, d_ft);
    const int nvec2 = H * D / v2;
    if (nvec2 >= 256 && nvec2 <= GAT_WAVES * 64 * SLICE_NI) {
        ProfScope prof2(v2 == 4 ? "gat_bwd_node_split_kernel<4>" : (v2 == 2 ? "gat_bwd_node_split_kernel<2>" : "gat_bwd_node_split_kernel<1>"),
                        s, 4.0 * (2.0 * n_nodes * (double)H * D + n_nodes * H), 1);
#define TXE_L(V)                                                                                                          \
    hipLaunchKernelGGL((gat_bwd_node_split_kernel<V>), dim3(nb), dim3(GAT_WAVES * 64), 0, s, rowptr_out, col_dst, pos_out, \
                       n_nodes, alpha, (const float*)dz_ws, H, D, attn_drop_p, scale, seed, d_pre, ld_dpre, d_ft, ld_dft,  \
                       d_a_src, ld_da)
        if (v2 == 4) TXE_L(4); else if (v2 == 2) TXE_L(2); else TXE_L(1);
#undef TXE_L
    } else {
        const int ni2 = pick_ni(nvec2);
        const KName kn("gat_bwd_node_kernel", v2, ni2);
        ProfScope prof2(kn.s, s, 4.0 * (2.0 * n_nodes * (double)H * D + n_nodes * H), 1);          // read d_pre + write d_ft
#define TXE_L(V, I)                                                                                                        \
    hipLaunchKernelGGL((gat_bwd_node_kernel<V, I>), dim3(nb), dim3(GAT_WAVES * 64), 0, s, rowptr_out, col_dst, pos_out,    \
                       n_nodes, alpha, (const float*)dz_ws, H, D, attn_drop_p, scale, seed, d_pre, ld_dpre, d_ft, ld_dft,  \
                       d_a_src, ld_da)
        TXE_DISPATCH_VEC_NI(v2, ni2, TXE_L);
#undef TXE_L
    }
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_leaky_relu_bwd(const float* d_out, const float* out_act, float slope, long long n, float* d_pre, void* stream) {
    if (n < 0 || (n > 0 && (!d_out || !out_act || !d_pre))) return TXE_ERR_ARG;
    if (n == 0) return TXE_OK;
    const int nb = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d_out, out_act, slope, n, d_pre);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_head_mean_fwd(const float* x, int H, int D, long long n_rows, float* y, void* stream) {
    if (H < 1 || D < 1 || n_rows < 0) return TXE_ERR_ARG;
    if (n_rows == 0) return TXE_OK;
    const long long total = n_rows * D;
    const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(head_mean_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, H, D, n_rows, y);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_head_mean_bwd(const float* dy, int H, int D, long long n_rows, float* dx, void* stream) {
    if (H < 1 || D < 1 || n_rows < 0) return TXE_ERR_ARG;
    if (n_rows == 0) return TXE_OK;
    const long long total = n_rows * H * D;
    const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(head_mean_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, H, D, n_rows, dx);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"
