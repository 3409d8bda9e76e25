// Adam / AMSGrad update of the whole parameter set in ONE launch (trainer.py:61 `self.optimizer.step()` with the optimizer of
// config.mag.json:66-73: Adam, lr 1e-3, weight_decay 0, amsgrad true).
// The model has 9 parameter tensors of very unequal sizes (3 ... 1,025,000 elements, 1.76 M in total); a multi-tensor apply that
// hands 64 K-element chunks to workgroups puts 27 workgroups on 256 CUs.  Here every workgroup owns 1,024 consecutive elements of
// one tensor (1,700 workgroups), 16-byte accesses: 9 streams x 7 MB = HBM-bound.
#include "txe_common.h"

#include <math.h>

namespace txe {

constexpr int ADAM_MAX_T = 24;          // tensors per launch (kernel-argument table)
constexpr int ADAM_CHUNK = 1024;        // elements per workgroup

struct AdamTable {
    float* p[ADAM_MAX_T];
    const float* g[ADAM_MAX_T];
    float* m[ADAM_MAX_T];
    float* v[ADAM_MAX_T];
    float* vmax[ADAM_MAX_T];
    long long n[ADAM_MAX_T];
    int first_chunk[ADAM_MAX_T + 1];
    int count;
};

struct AdamScalars {
    float one_minus_b1, b2, one_minus_b2, eps, wd, step_size, bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float& vm, bool ams, const AdamScalars& c) {
    g = fmaf(c.wd, p, g);                                   // wd == 0: g unchanged
    m = m + (g - m) * c.one_minus_b1;                       // lerp(m, g, 1 - beta1)
    v = c.b2 * v + c.one_minus_b2 * g * g;
    float d;
    if (ams) { vm = fmaxf(vm, v); d = sqrtf(vm) / c.bc2_sqrt + c.eps; }
    else d = sqrtf(v) / c.bc2_sqrt + c.eps;
    p -= c.step_size * m / d;
}

template <bool AMS>
__global__ __launch_bounds__(256) void adam_kernel(AdamTable T, AdamScalars c) {
    int t = 0;
#pragma unroll 1
    while (t + 1 < T.count && (int)blockIdx.x >= T.first_chunk[t + 1]) ++t;
    const long long n = T.n[t];
    const long long i0 = (long long)(blockIdx.x - T.first_chunk[t]) * ADAM_CHUNK + 4 * threadIdx.x;
    if (i0 >= n) return;
    float* __restrict__ P = T.p[t];
    const float* __restrict__ G = T.g[t];
    float* __restrict__ M = T.m[t];
    float* __restrict__ V = T.v[t];
    float* __restrict__ X = AMS ? T.vmax[t] : T.v[t];
    const bool vec = (i0 + 3 < n) && ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V | (uintptr_t)X) & 15) == 0);
    if (vec) {
        float4 p = *reinterpret_cast<const float4*>(P + i0);
        const float4 g = *reinterpret_cast<const float4*>(G + i0);
        float4 m = *reinterpret_cast<const float4*>(M + i0);
        float4 v = *reinterpret_cast<const float4*>(V + i0);
        float4 x = *reinterpret_cast<const float4*>(X + i0);
        adam_one(p.x, g.x, m.x, v.x, x.x, AMS, c);
        adam_one(p.y, g.y, m.y, v.y, x.y, AMS, c);
        adam_one(p.z, g.z, m.z, v.z, x.z, AMS, c);
        adam_one(p.w, g.w, m.w, v.w, x.w, AMS, c);
        *reinterpret_cast<float4*>(P + i0) = p;
        *reinterpret_cast<float4*>(M + i0) = m;
        *reinterpret_cast<float4*>(V + i0) = v;
        if (AMS) *reinterpret_cast<float4*>(X + i0) = x;
    } else {
        for (long long i = i0; i < n && i < i0 + 4; ++i) {
            float p = P[i], m = M[i], v = V[i], x = AMS ? X[i] : 0.f;
            adam_one(p, G[i], m, v, x, AMS, c);
            P[i] = p; M[i] = m; V[i] = v;
            if (AMS) X[i] = x;
        }
    }
}

}  // namespace txe

using namespace txe;

extern "C" {

// One optimizer step over n_tensors parameter tensors (HOST arrays of DEVICE pointers; numel[t] elements each, fp32, dense).
//   g' = g + weight_decay p;  m = lerp(m, g', 1-beta1);  v = beta2 v + (1-beta2) g'^2;  [vmax = max(vmax, v)]
//   p -= lr / (1 - beta1^step) * m / (sqrt(vmax or v) / sqrt(1 - beta2^step) + eps)
// `step` is the 1-based count of THIS update (shared by the tensors of the call); max_exp_avg_sq == NULL selects plain Adam.
int txe_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  float* const* max_exp_avg_sq, const long long* numel, double lr, double beta1, double beta2, double eps,
                  double weight_decay, long long step, void* stream) {
    if (n_tensors < 0 || step < 1 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel))) return TXE_ERR_ARG;
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0)) return TXE_ERR_ARG;
    AdamScalars c;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    c.one_minus_b1 = (float)(1.0 - beta1);
    c.b2 = (float)beta2;
    c.one_minus_b2 = (float)(1.0 - beta2);
    c.eps = (float)eps;
    c.wd = (float)weight_decay;
    c.step_size = (float)(lr / bc1);
    c.bc2_sqrt = (float)sqrt(bc2);
    const bool ams = max_exp_avg_sq != nullptr;
    for (int t = 0; t < n_tensors;) {                  // one launch per ADAM_MAX_T non-empty tensors
        AdamTable T;
        T.count = 0;
        long long chunks = 0, elems = 0;
        for (; t < n_tensors && T.count < ADAM_MAX_T; ++t) {
            if (numel[t] < 0) return TXE_ERR_ARG;
            if (numel[t] == 0) continue;
            if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t] || (ams && !max_exp_avg_sq[t])) return TXE_ERR_ARG;
            const int k = T.count++;
            T.p[k] = params[t]; T.g[k] = grads[t]; T.m[k] = exp_avg[t]; T.v[k] = exp_avg_sq[t];
            T.vmax[k] = ams ? max_exp_avg_sq[t] : exp_avg_sq[t];
            T.n[k] = numel[t];
            elems += numel[t];
            T.first_chunk[k] = (int)chunks;
            chunks += (numel[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (chunks > 0x7fffffffLL) return TXE_ERR_ARG;
        }
        if (T.count == 0) continue;
        T.first_chunk[T.count] = (int)chunks;
        for (int k = T.count; k < ADAM_MAX_T; ++k) { T.p[k] = nullptr; T.g[k] = nullptr; T.m[k] = nullptr; T.v[k] = nullptr; T.vmax[k] = nullptr; T.n[k] = 0; T.first_chunk[k + 1] = (int)chunks; }
        ProfScope prof(ams ? "adam_kernel<true>" : "adam_kernel<false>", (hipStream_t)stream, 4.0 * elems * (ams ? 9 : 7), 1);   // p, g, m, v [, vmax] read; all but g written
        if (ams) hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, T, c);
        else hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, T, c);
        TXE_CHECK_LAUNCH();
    }
    return TXE_OK;
}

}  // extern "C"
