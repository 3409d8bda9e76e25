// Deterministic two-stage column sum  out[j] = sum_m x[m][j]  (bias gradients: model_zoo.py:47 `h + self.bias`).
// stage 1: workgroup (column tile of 64, chunk of COLSUM_ROWS rows): 64 column lanes x 4 row groups, 8 independent loads in flight per
//          thread, fixed-order LDS combine -> part[chunk][j];   stage 2: the chunks, same shape.
// (First version: one thread walked 256 rows of a column with one load in flight -- 244 us for an 18 k x 500 matrix, 40 % of
//  the PGCN training step.)
#pragma once
#include "txe_common.h"

namespace txe {

constexpr int COLSUM_ROWS = 128;

static inline int colsum_chunks(long long n_rows) { return (int)((n_rows + COLSUM_ROWS - 1) / COLSUM_ROWS); }
static inline size_t colsum_ws_bytes(long long n_rows, int cols) {
    const int c = colsum_chunks(n_rows);
    return (size_t)(c > 0 ? c : 1) * cols * sizeof(float);
}

// x rows [r0, r1) of the chunk blockIdx.y; rows == the chunk count and chunk size 1 turn it into stage 2
static __global__ __launch_bounds__(256) void colsum_chunk_kernel(const float* __restrict__ x, long long ldx, int n_rows, int cols,
                                                                  int rows_per_chunk, float* __restrict__ out /*[chunks][cols]*/) {
    __shared__ float red[4][64];
    const int jl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const int jc = (j < cols) ? j : 0;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(n_rows, r0 + rows_per_chunk);
    float acc = 0.f;
    for (int m0 = r0 + rg; m0 < r1; m0 += 32) {                 // 8 rows of this row group per step, clamped and weighted
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[(long long)min(m0 + 4 * q, r1 - 1) * ldx + jc];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += (m0 + 4 * q < r1) ? v[q] : 0.f;
    }
    red[rg][jl] = acc;
    __syncthreads();
    if (rg == 0 && j < cols) out[(long long)blockIdx.y * cols + j] = red[0][jl] + red[1][jl] + red[2][jl] + red[3][jl];
}

// out [cols] = column sums of x [n_rows][cols] (row stride ldx); part: colsum_ws_bytes(n_rows, cols) of scratch.  n_rows == 0 -> zeros.
static inline int colsum_launch(const float* x, long long ldx, long long n_rows, int cols, float* part, float* out, hipStream_t s) {
    const int chunks = colsum_chunks(n_rows);
    const dim3 g1((cols + 63) / 64, chunks > 0 ? chunks : 1);
    if (chunks > 0) {
        hipLaunchKernelGGL(colsum_chunk_kernel, g1, dim3(256), 0, s, x, ldx, (int)n_rows, cols, COLSUM_ROWS, part);
        TXE_CHECK_LAUNCH();
    }
    // stage 2: one "chunk" holding all partial rows (chunks == 0: sums nothing, writes zeros)
    hipLaunchKernelGGL(colsum_chunk_kernel, dim3((cols + 63) / 64, 1), dim3(256), 0, s, (const float*)part, (long long)cols, chunks, cols,
                       chunks > 0 ? chunks : 1, out);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // namespace txe
