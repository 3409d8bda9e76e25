// InfoNCE training loss (model/loss.py:52-57: F.cross_entropy(output, target, reduction="sum") on the [queries][1 + negatives]
// regrouping of trainer.py:52-56) with its gradient, in one launch: torch runs log-softmax, nll, their two backward kernels and
// three fills for it -- seven ~5 us dispatches around 4,096 numbers.
#include "txe_common.h"

namespace txe {

constexpr int NCE_WAVES = 16;

// one workgroup; wave w owns rows w, w+16, ...; the row losses are combined in a fixed order (deterministic).
__global__ __launch_bounds__(NCE_WAVES * 64) void info_nce_kernel(const float* __restrict__ x, long long ld_x, int B, int Cc,
                                                                  const long long* __restrict__ target, float* __restrict__ loss,
                                                                  float* __restrict__ d_x, long long ld_dx) {
    __shared__ float s_part[NCE_WAVES];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    float acc = 0.f;                                   // lane 0 of every wave: sum of its rows' losses, in row order
    for (int b = w; b < B; b += NCE_WAVES) {
        const float* row = x + (long long)b * ld_x;
        const int t = target ? (int)target[b] : 0;
        float m = -INFINITY;
        for (int c = l; c < Cc; c += 64) m = fmaxf(m, row[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = l; c < Cc; c += 64) s += __expf(row[c] - m);
        s = wave_sum(s);
        const float lse = m + __logf(s);
        const float inv = 1.f / s;
        float* drow = d_x + (long long)b * ld_dx;
        for (int c = l; c < Cc; c += 64) drow[c] = __expf(row[c] - m) * inv - (c == t ? 1.f : 0.f);
        if (l == 0) acc += lse - row[t];
    }
    if (l == 0) s_part[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < NCE_WAVES; ++i) tot += s_part[i];
        loss[0] = tot;
    }
}

}  // namespace txe

using namespace txe;

extern "C" {

// loss[0] = sum_b (logsumexp(x[b][:]) - x[b][target[b]]);  d_x[b][c] = softmax(x[b])[c] - [c == target[b]]
// target == NULL: every row's positive is column 0 (trainer.py:53 builds exactly that).  0 <= target[b] < C is the caller's contract.
int txe_info_nce(const float* x, long long ld_x, int B, int Cc, const long long* target, float* loss, float* d_x, long long ld_dx,
                 void* stream) {
    if (B < 0 || Cc < 1 || ld_x < Cc || ld_dx < Cc || !loss || (B > 0 && (!x || !d_x))) return TXE_ERR_ARG;
    hipLaunchKernelGGL(info_nce_kernel, dim3(1), dim3(NCE_WAVES * 64), 0, (hipStream_t)stream, x, ld_x, B, Cc, target, loss, d_x, ld_dx);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"
