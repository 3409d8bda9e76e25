// Egonet construction and batching ON DEVICE: the per-anchor Python of data_loader/dataset.py:404-437 plus
// dgl.batch (data_loaders.py:25), straight from the taxonomy's parent / child CSR to the batched graph's node table and
// BOTH CSR views -- no per-egonet objects, no sort: the egonet's edge list has a closed form.
//
// Egonet i of anchor a with k parents and m kept children (n = k+1+m nodes, 2n-1 edges, dataset.py:429-435):
//   local nodes      : 0..k-1 grand-parents (pos 0), k anchor (pos 1), k+1.. siblings (pos 2)
//   edge ids         : [0,k) gp_j -> anchor | [k,k+m) anchor -> sib_j | [k+m, k+m+n) self loops
//   in-edges  (dst)  : gp_j: {self} | anchor: {gp_0..gp_{k-1}, self} | sib_j: {anchor, self}          (edge-id order)
//   out-edges (src)  : gp_j: {->anchor, self} | anchor: {->sib_0.., self} | sib_j: {self}
// Children beyond expand_factor are drawn WITH replacement (random.choices, dataset.py:419) from a counter-based hash;
// exclude[i] >= 0 drops that query node from the sibling set (the positive example, instance_mode 1, dataset.py:421-424).
#include <hipcub/hipcub.hpp>

#include "txe_common.h"

namespace txe {

__device__ __forceinline__ int draw_child(unsigned long long seed, int i, int t, int deg) {
    return (int)(mix64(seed ^ mix64(((unsigned long long)i << 20) + (unsigned long long)t)) % (unsigned long long)deg);
}

// sizes: n[i] = k + 1 + m.  One wavefront per egonet, lane t = sibling slot t: the (up to expand_factor) child look-ups of an egonet
// are independent loads of one wave instead of a 50-deep dependent chain of one thread (55-75 us per 4,096-egonet batch before).
__global__ __launch_bounds__(256) void egonet_sizes_kernel(const int* __restrict__ par_ptr, const int* __restrict__ chd_ptr,
                                                           const int* __restrict__ chd_idx, const int* __restrict__ anchors,
                                                           const int* __restrict__ exclude, int G, int expand, unsigned long long seed,
                                                           int index_base, int* __restrict__ n_nodes) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + w;
    if (i >= G) return;
    const int a = anchors[i];
    const int k = par_ptr[a + 1] - par_ptr[a];
    const int cb = chd_ptr[a], deg = chd_ptr[a + 1] - cb;
    const int ex = exclude ? exclude[i] : -1;
    const bool all = deg <= expand;
    const int draws = all ? deg : expand;
    int m = 0;
    if (all && ex < 0) {
        m = deg;
    } else {
        for (int t0 = 0; t0 < draws; t0 += 64) {
            const int t = t0 + l;
            const bool live = t < draws;
            const int c = live ? chd_idx[cb + (all ? t : draw_child(seed, index_base + i, t, deg))] : ex;
            m += __popcll(__ballot(live && c != ex));
        }
    }
    if (l == 0) n_nodes[i] = k + 1 + m;
}

// one wavefront per egonet: node table + both CSR views
__global__ __launch_bounds__(256) void egonet_fill_kernel(const int* __restrict__ par_ptr, const int* __restrict__ par_idx,
                                                          const int* __restrict__ chd_ptr, const int* __restrict__ chd_idx,
                                                          const int* __restrict__ anchors, const int* __restrict__ exclude, int G,
                                                          int expand, unsigned long long seed, int index_base, const int* __restrict__ node_off,
                                                          int* __restrict__ ids, int* __restrict__ pos, int* __restrict__ rowptr_in,
                                                          int* __restrict__ col_src, int* __restrict__ eid_in,
                                                          int* __restrict__ rowptr_out, int* __restrict__ col_dst,
                                                          int* __restrict__ pos_out) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + w;
    if (i >= G) return;
    const int a = anchors[i];
    const int pb = par_ptr[a], k = par_ptr[a + 1] - pb;
    const int cb = chd_ptr[a], deg = chd_ptr[a + 1] - cb;
    const int ex = exclude ? exclude[i] : -1;
    const int n0 = node_off[i], n = node_off[i + 1] - n0;
    const int m = n - k - 1;
    const int e0 = 2 * n0 - i;                        // edges of the egonets before this one: sum (2 n_j - 1)
    // ---- node table ----
    for (int j = l; j < k; j += 64) { ids[n0 + j] = par_idx[pb + j]; pos[n0 + j] = 0; }
    if (l == 0) { ids[n0 + k] = a; pos[n0 + k] = 1; }
    {                                                 // siblings keep their order: lane t = slot t, compaction by ballot + prefix count
        int o = n0 + k + 1;
        const bool all = deg <= expand;
        const int draws = all ? deg : expand;
        for (int t0 = 0; t0 < draws; t0 += 64) {
            const int t = t0 + l;
            const bool live = t < draws;
            const int c = live ? chd_idx[cb + (all ? t : draw_child(seed, index_base + i, t, deg))] : ex;
            const bool keep = live && c != ex;
            const unsigned long long mk = __ballot(keep);
            if (keep) {
                const int dst = o + __popcll(mk & ((1ull << l) - 1ull));
                ids[dst] = c; pos[dst] = 2;
            }
            o += __popcll(mk);
        }
    }
    // ---- destination-sorted CSR (positions relative to e0) ----
    //   gp_j: p = j (self) | anchor: p = k + {0..k-1} (parents), 2k (self) | sib_j: p = 2k+1+2j (anchor), 2k+2+2j (self)
    const int self0 = k + m;                          // local edge id of node 0's self loop
    for (int j = l; j < k; j += 64) {
        rowptr_in[n0 + j] = e0 + j;
        col_src[e0 + j] = n0 + j;            eid_in[e0 + j] = e0 + self0 + j;
        col_src[e0 + k + j] = n0 + j;        eid_in[e0 + k + j] = e0 + j;
    }
    if (l == 0) {
        rowptr_in[n0 + k] = e0 + k;
        col_src[e0 + 2 * k] = n0 + k;        eid_in[e0 + 2 * k] = e0 + self0 + k;
    }
    for (int j = l; j < m; j += 64) {
        const int p = e0 + 2 * k + 1 + 2 * j;
        rowptr_in[n0 + k + 1 + j] = p;
        col_src[p] = n0 + k;                 eid_in[p] = e0 + k + j;
        col_src[p + 1] = n0 + k + 1 + j;     eid_in[p + 1] = e0 + self0 + k + 1 + j;
    }
    // ---- source-sorted CSR ----
    //   gp_j: q = 2j (->anchor), 2j+1 (self) | anchor: q = 2k + {0..m-1} (->sib), 2k+m (self) | sib_j: q = 2k+m+1+j (self)
    for (int j = l; j < k; j += 64) {
        const int q = e0 + 2 * j;
        rowptr_out[n0 + j] = q;
        col_dst[q] = n0 + k;                 pos_out[q] = e0 + k + j;
        col_dst[q + 1] = n0 + j;             pos_out[q + 1] = e0 + j;
    }
    if (l == 0) {
        rowptr_out[n0 + k] = e0 + 2 * k;
        col_dst[e0 + 2 * k + m] = n0 + k;    pos_out[e0 + 2 * k + m] = e0 + 2 * k;
    }
    for (int j = l; j < m; j += 64) {
        col_dst[e0 + 2 * k + j] = n0 + k + 1 + j;      pos_out[e0 + 2 * k + j] = e0 + 2 * k + 1 + 2 * j;
        const int q = e0 + 2 * k + m + 1 + j;
        rowptr_out[n0 + k + 1 + j] = q;
        col_dst[q] = n0 + k + 1 + j;                   pos_out[q] = e0 + 2 * k + 2 + 2 * j;
    }
    if (i == G - 1 && l == 0) { rowptr_in[n0 + n] = e0 + 2 * n - 1; rowptr_out[n0 + n] = e0 + 2 * n - 1; }
}

static size_t scan_temp_bytes(int G) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int*)nullptr, (int*)nullptr, G + 1);
    return bytes;
}

}  // namespace txe

using namespace txe;

extern "C" {

size_t txe_egonet_ws_bytes(int G) { return ((size_t)(G + 1) * 4 + 255) / 256 * 256 + scan_temp_bytes(G); }

// Step 1: node_off [G+1] (exclusive prefix sum of the egonet sizes; node_off[G] = total nodes N; total edges = 2N - G).
// anchors [G], exclude [G] or NULL (query node to drop from each egonet's siblings, -1 = none).
int txe_egonet_offsets(const int* par_ptr, const int* chd_ptr, const int* chd_idx, const int* anchors, const int* exclude, int G,
                       int expand, unsigned long long seed, int index_base, int* node_off, void* ws, size_t ws_bytes, void* stream) {
    if (G < 0 || expand < 0 || !par_ptr || !chd_ptr || !chd_idx || !node_off || !ws || (G > 0 && !anchors)) return TXE_ERR_ARG;
    if (ws_bytes < txe_egonet_ws_bytes(G)) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int* sizes = (int*)ws;
    void* temp = (char*)ws + ((size_t)(G + 1) * 4 + 255) / 256 * 256;
    size_t temp_bytes = scan_temp_bytes(G);
    (void)hipMemsetAsync(sizes + G, 0, 4, s);
    if (G > 0) {
        hipLaunchKernelGGL(egonet_sizes_kernel, dim3((G + 3) / 4), dim3(256), 0, s, par_ptr, chd_ptr, chd_idx, anchors, exclude, G, expand,
                           seed, index_base, sizes);
        TXE_CHECK_LAUNCH();
    }
    if (hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, (const int*)sizes, node_off, G + 1, s) != hipSuccess) return TXE_ERR_LAUNCH;
    return TXE_OK;
}

// Step 2: node table (ids, pos [N]) and both CSR views (rowptr_* [N+1], col_* / eid_in / pos_out [2N-G]) of the batch.
int txe_egonet_fill(const int* par_ptr, const int* par_idx, const int* chd_ptr, const int* chd_idx, const int* anchors,
                    const int* exclude, int G, int expand, unsigned long long seed, int index_base, const int* node_off, int* ids, int* pos,
                    int* rowptr_in, int* col_src, int* eid_in, int* rowptr_out, int* col_dst, int* pos_out, void* stream) {
    if (G < 0 || !par_ptr || !par_idx || !chd_ptr || !chd_idx || !node_off || !rowptr_in || !rowptr_out) return TXE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (G == 0) {
        (void)hipMemsetAsync(rowptr_in, 0, 4, s);
        (void)hipMemsetAsync(rowptr_out, 0, 4, s);
        return TXE_OK;
    }
    if (!anchors || !ids || !pos || !col_src || !eid_in || !col_dst || !pos_out) return TXE_ERR_ARG;
    hipLaunchKernelGGL(egonet_fill_kernel, dim3((G + 3) / 4), dim3(256), 0, s, par_ptr, par_idx, chd_ptr, chd_idx, anchors, exclude, G, expand,
                       seed, index_base, node_off, ids, pos, rowptr_in, col_src, eid_in, rowptr_out, col_dst, pos_out);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"
