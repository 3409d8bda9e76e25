// Device-side construction of the two CSR views of a batched egonet graph from its COO edge list
// (the structure dgl.batch produces, data_loaders.py:25; edge order of dataset.py:431-435).
//   destination-sorted: rowptr_in / col_src / eid_in   (stable: in-edges of a node stay in edge-id order, which
//                       fixes the floating-point summation order of the aggregation and makes runs repeatable)
//   source-sorted:      rowptr_out / col_dst / pos_out  (pos_out = index of the edge in destination order)
// The two stable sorts use rocPRIM's radix sort through hipCUB (library code: this is one-off index preparation,
// not part of the per-step hot path); everything else is hand written.
#include <hipcub/hipcub.hpp>

#include "txe_common.h"

namespace txe {

__global__ void iota_kernel(int* x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = i;
}

// rowptr[v] = first index p with sorted_keys[p] >= v   (v = 0..n_nodes)
__global__ void lower_bound_kernel(const int* __restrict__ sorted_keys, int n_edges, int n_nodes, int* __restrict__ rowptr) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n_nodes) return;
    int lo = 0, hi = n_edges;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorted_keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    rowptr[v] = lo;
}

__global__ void gather_in_kernel(const int* __restrict__ order_in, const int* __restrict__ src, int n_edges,
                                 int* __restrict__ col_src, int* __restrict__ inv_in) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_edges) return;
    const int e = order_in[p];
    col_src[p] = src[e];
    inv_in[e] = p;
}

__global__ void gather_out_kernel(const int* __restrict__ order_out, const int* __restrict__ dst, const int* __restrict__ inv_in,
                                  int n_edges, int* __restrict__ col_dst, int* __restrict__ pos_out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_edges) return;
    const int e = order_out[j];
    col_dst[j] = dst[e];
    pos_out[j] = inv_in[e];
}

static inline size_t g_align(size_t x) { return (x + 255) / 256 * 256; }

static size_t sort_temp_bytes(int n_edges) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr,
                                       n_edges > 0 ? n_edges : 1);
    return bytes;
}

}  // namespace txe

using namespace txe;

extern "C" {

// host evaluation of the dropout hash (uniform in [0,1) for (seed, index)); lets tests pin taxoexpan_amd/rng.py to the
// very function the kernels inline.
float txe_dropout_uniform_host(unsigned long long seed, unsigned long long idx) { return txe::uniform01(seed, idx); }
unsigned txe_dropout_mask_word_host(unsigned long long seed, unsigned long long word_index, float p) {
    return txe::drop_mask_word(seed, word_index, (unsigned)(p * 65536.0f + 0.5f));
}

size_t txe_build_csr_ws_bytes(int n_nodes, int n_edges) {
    (void)n_nodes;
    const size_t e = g_align((size_t)(n_edges > 0 ? n_edges : 1) * 4);
    return 4 * e + g_align(sort_temp_bytes(n_edges));
}

// src/dst [E] int32 COO (edge-id order).  Outputs: rowptr_in [N+1], col_src [E], eid_in [E], rowptr_out [N+1],
// col_dst [E], pos_out [E].
int txe_build_csr(const int* src, const int* dst, int n_nodes, int n_edges, int* rowptr_in, int* col_src, int* eid_in,
                  int* rowptr_out, int* col_dst, int* pos_out, void* ws, size_t ws_bytes, void* stream) {
    if (n_nodes < 0 || n_edges < 0 || !rowptr_in || !rowptr_out || !ws) return TXE_ERR_ARG;
    if (n_edges > 0 && (!src || !dst || !col_src || !eid_in || !col_dst || !pos_out)) return TXE_ERR_ARG;
    if (ws_bytes < txe_build_csr_ws_bytes(n_nodes, n_edges)) return TXE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t e = g_align((size_t)(n_edges > 0 ? n_edges : 1) * 4);
    char* b = (char*)ws;
    int* eid = (int*)b;
    int* keys_sorted = (int*)(b + e);
    int* order_out = (int*)(b + 2 * e);
    int* inv_in = (int*)(b + 3 * e);
    void* temp = b + 4 * e;
    size_t temp_bytes = sort_temp_bytes(n_edges);
    const int nbn = (n_nodes + 1 + 255) / 256;
    if (n_edges == 0) {
        hipLaunchKernelGGL(lower_bound_kernel, dim3(nbn), dim3(256), 0, s, (const int*)nullptr, 0, n_nodes, rowptr_in);
        hipLaunchKernelGGL(lower_bound_kernel, dim3(nbn), dim3(256), 0, s, (const int*)nullptr, 0, n_nodes, rowptr_out);
        TXE_CHECK_LAUNCH();
        return TXE_OK;
    }
    const int nbe = (n_edges + 255) / 256;
    int bits = 1;
    while ((1ll << bits) < (long long)n_nodes + 1) ++bits;
    hipLaunchKernelGGL(iota_kernel, dim3(nbe), dim3(256), 0, s, eid, n_edges);
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, dst, keys_sorted, (const int*)eid, eid_in, n_edges, 0, bits, s) != hipSuccess)
        return TXE_ERR_LAUNCH;
    hipLaunchKernelGGL(lower_bound_kernel, dim3(nbn), dim3(256), 0, s, (const int*)keys_sorted, n_edges, n_nodes, rowptr_in);
    hipLaunchKernelGGL(gather_in_kernel, dim3(nbe), dim3(256), 0, s, (const int*)eid_in, src, n_edges, col_src, inv_in);
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, src, keys_sorted, (const int*)eid, order_out, n_edges, 0, bits, s) != hipSuccess)
        return TXE_ERR_LAUNCH;
    hipLaunchKernelGGL(lower_bound_kernel, dim3(nbn), dim3(256), 0, s, (const int*)keys_sorted, n_edges, n_nodes, rowptr_out);
    hipLaunchKernelGGL(gather_out_kernel, dim3(nbe), dim3(256), 0, s, (const int*)order_out, dst, (const int*)inv_in, n_edges,
                       col_dst, pos_out);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"
