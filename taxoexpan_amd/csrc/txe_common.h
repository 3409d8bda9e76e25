// Shared device/host helpers for the TaxoExpan MI355X (gfx950) kernels.
// Wavefront = 64 lanes everywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TXE_OK 0
#define TXE_ERR_ARG -1
#define TXE_ERR_LAUNCH -2
#define TXE_ERR_WORKSPACE -3
#define TXE_TAIL_CHAIN_BYTES 1024     /* include/txe.h */

#define TXE_WAVE 64
#define TXE_NUM_XCD 8

#define TXE_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return TXE_ERR_LAUNCH;       \
    } while (0)

namespace txe {

// MI355X dispatches workgroup b to XCD (b % 8); each XCD has a private 4 MiB L2.  Remap the
// hardware block id so that every XCD works on ONE contiguous range of logical blocks: neighbouring
// logical blocks (nodes of the same egonet, tiles that share an operand panel) then share an L2.
// Bijective for every nblocks (MI355X_MICROARCH.md "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks / TXE_NUM_XCD, r = nblocks % TXE_NUM_XCD;
    const int xcd = bid % TXE_NUM_XCD, idx = bid / TXE_NUM_XCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Wavefront reductions on the DPP cross-lane network (VALU operands, a few cycles each) instead of __shfl_xor, which hipcc lowers to
// ds_bpermute_b32 -- six dependent trips through the LDS pipeline (~100+ cycles each) per reduction, and the sweeps do one per edge.
// Within a row of 16 lanes: two quad permutes, row_half_mirror, row_mirror leave the row total in every lane; the four row totals
// are then combined through v_readlane.  Every lane returns the total.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);       // row_half_mirror
    v += dpp_f<0x140>(v);       // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Counter-based dropout: keep(seed, idx) is a pure function, so forward, backward and the host-side
// test restatement (taxoexpan_amd/rng.py) regenerate the same mask without storing it.
// splitmix64 finaliser; the top 24 bits give a uniform in [0,1).
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
    const uint64_t h = mix64(seed ^ mix64(idx));
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}
// returns the multiplicative factor: 0 (dropped) or 1/(1-p) (kept); p == 0 -> 1.
__host__ __device__ __forceinline__ float drop_factor(uint64_t seed, uint64_t idx, float p, float scale) {
    return (uniform01(seed, idx) >= p) ? scale : 0.0f;
}

// Feature-dropout bit mask (1 = keep), 32 columns per word, words_per_row = ceil(cols/32).  Bit b of word w is
// [u(w, b) >= thr16], thr16 = round(p * 65536), where the 16-bit uniform u(w, b) is assembled BIT-PLANE-WISE: its bit j (15 = most
// significant) is bit b of the plane word R(w, j) = 32-bit half (j & 1) of mix64(seed + (8w + (j >> 1)) * W) -- a SplitMix64 stream with
// a Weyl step W per counter.  The comparison then runs on all 32 columns of a word at once,
//     ge = ~0;   for j = (lowest set bit of thr16) .. 15:   ge = (thr16 >> j) & 1 ? ge & R_j : ge | R_j
// and the planes below thr16's lowest set bit never matter: p = 0.5 (thr16 = 0x8000, PGAT's default) costs ONE finaliser per mask
// word, p = 0.1 eight -- against eight for every p when each column compared a 16-bit chunk of its own (the mask of the folded
// layer's 2,080-column input is 1.2 M words per step: 12 us of VALU work across the whole chip).
__host__ __device__ __forceinline__ unsigned drop_mask_word(uint64_t seed, uint64_t word_index, unsigned thr16) {
    if (thr16 == 0u) return 0xFFFFFFFFu;
    if (thr16 > 0xFFFFu) return 0u;
    int j0 = 0;
    while (((thr16 >> j0) & 1u) == 0u) ++j0;
    unsigned ge = 0xFFFFFFFFu;
    for (int k = j0 >> 1; k < 8; ++k) {
        const uint64_t h = mix64(seed + (word_index * 8 + k) * 0xD1342543DE82EF95ull);
        for (int half = 0; half < 2; ++half) {
            const int j = 2 * k + half;
            if (j < j0) continue;
            const unsigned r = (unsigned)(h >> (32 * half));
            ge = ((thr16 >> j) & 1u) ? (ge & r) : (ge | r);
        }
    }
    return ge;
}

__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }

// ---- best-k lists (top-k mode of the scoring loop) ----------------------------------------------------------------------
// A list of TOPK_MAX (key, column) pairs in registers, best first.  "better": larger key, equal keys by smaller column -- the order
// of Python's stable `sorted(enumerate(scores), key=lambda x: -x[1])` (infer.py:100, test_fast.py:125).  Empty slots hold
// (-inf, INT_MAX): worse than any real entry, a real -inf key included.
constexpr int TOPK_MAX = 8;
__device__ __forceinline__ bool topk_better(float ka, int ia, float kb, int ib) { return (ka > kb) | ((ka == kb) & (ia < ib)); }
__device__ __forceinline__ void topk_init(float* bk, int* bi) {
#pragma unroll
    for (int t = 0; t < TOPK_MAX; ++t) { bk[t] = -INFINITY; bi[t] = 0x7fffffff; }
}
// (key, idx) takes the place of the last entry and bubbles up: all indices static, the list never leaves its registers
__device__ __forceinline__ void topk_insert(float* bk, int* bi, float key, int idx) {
    bk[TOPK_MAX - 1] = key; bi[TOPK_MAX - 1] = idx;
#pragma unroll
    for (int t = TOPK_MAX - 1; t > 0; --t) {
        const bool up = topk_better(bk[t], bi[t], bk[t - 1], bi[t - 1]);
        const float k0 = bk[t - 1]; const int i0 = bi[t - 1];
        bk[t - 1] = up ? bk[t] : k0; bi[t - 1] = up ? bi[t] : i0;
        bk[t] = up ? k0 : bk[t];     bi[t] = up ? i0 : bi[t];
    }
}
// float keys as ints of the same order (for atomicMax): non-negative floats keep their bits, negative ones flip their magnitude
__device__ __forceinline__ int topk_ord(float x) { const int i = __float_as_int(x); return i ^ ((i >> 31) & 0x7fffffff); }
__device__ __forceinline__ float topk_unord(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7fffffff)); }
// entry k-1 of a list (k uniform, 1 <= k <= TOPK_MAX): the current k-th best -- what a candidate has to beat when only k entries count
__device__ __forceinline__ void topk_kth(const float* bk, const int* bi, int k, float& wk, int& wi) {
    wk = bk[0]; wi = bi[0];
#pragma unroll
    for (int t = 1; t < TOPK_MAX; ++t) { wk = (t < k) ? bk[t] : wk; wi = (t < k) ? bi[t] : wi; }
}
__device__ __forceinline__ float topk_key_of(float x, bool larger) {
    const float k = larger ? x : -x;
    return (k != k) ? -INFINITY : k;                 // NaN compares false with everything: it ranks last
}


// ---- optional per-launch timing (txe_profile.hip); a no-op unless txe_profile_enable(1) was called ----
bool prof_enabled();
int prof_begin(const char* name, hipStream_t s, double work, int kind);
void prof_end(int id, hipStream_t s);
struct ProfScope {
    int id;
    hipStream_t s;
    ProfScope(const char* name, hipStream_t st, double work, int kind) : id(-1), s(st) {
        if (prof_enabled()) id = prof_begin(name, st, work, kind);
    }
    ~ProfScope() {
        if (id >= 0) prof_end(id, s);
    }
};

}  // namespace txe
