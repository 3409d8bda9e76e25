// InfoNCE training loss (model/loss.py:52-57: F.cross_entropy(output, target, reduction="sum") on the [queries][1 + negatives]
// regrouping of trainer.py:52-56) with its gradient, in one launch: torch runs log-softmax, nll, their two backward kernels and
// three fills for it -- seven ~5 us dispatches around 4,096 numbers.
#include "txe_common.h"

namespace txe {

constexpr int NCE_WAVES = 16;

// One workgroup.  A row occupies a group of GW lanes (GW = 32 when C <= 32, else 64; longer rows loop), so a wave handles 64/GW rows
// per pass, and the passes of a wave are independent loads issued back to back.  The row losses are combined in a fixed order.
template <int GW>
__global__ __launch_bounds__(NCE_WAVES * 64) void info_nce_kernel(const float* __restrict__ x, long long ld_x, int B, int Cc,
                                                                  const long long* __restrict__ target, float* __restrict__ loss,
                                                                  float* __restrict__ d_x, long long ld_dx) {
    constexpr int RPW = 64 / GW;                       // rows per wave and pass
    __shared__ float s_part[NCE_WAVES * RPW];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int sub = l / GW, gl = l % GW;
    float acc = 0.f;                                   // lane 0 of every group: sum of its rows' losses, in row order
    for (int b0 = 0; b0 < B; b0 += NCE_WAVES * RPW) {
        const int b = b0 + w * RPW + sub;
        const bool live = b < B;
        const float* row = x + (long long)(live ? b : 0) * ld_x;
        const int t = (live && target) ? (int)target[b] : 0;
        float m = -INFINITY;
        for (int c = gl; c < Cc; c += GW) m = fmaxf(m, row[c]);
#pragma unroll
        for (int o = GW / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f;
        for (int c = gl; c < Cc; c += GW) s += __expf(row[c] - m);
#pragma unroll
        for (int o = GW / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float inv = 1.f / s;
        if (live) {
            float* drow = d_x + (long long)b * ld_dx;
            for (int c = gl; c < Cc; c += GW) drow[c] = __expf(row[c] - m) * inv - (c == t ? 1.f : 0.f);
            if (gl == 0) acc += m + __logf(s) - row[t];
        }
    }
    if (gl == 0) s_part[w * RPW + sub] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < NCE_WAVES * RPW; ++i) tot += s_part[i];
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
    if (Cc <= 32) hipLaunchKernelGGL(info_nce_kernel<32>, dim3(1), dim3(NCE_WAVES * 64), 0, (hipStream_t)stream, x, ld_x, B, Cc, target, loss, d_x, ld_dx);
    else hipLaunchKernelGGL(info_nce_kernel<64>, dim3(1), dim3(NCE_WAVES * 64), 0, (hipStream_t)stream, x, ld_x, B, Cc, target, loss, d_x, ld_dx);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // extern "C"
