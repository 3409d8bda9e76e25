// C = A B^T in fp32 accuracy on the bf16 matrix pipe (three-plane operands, six plane products: txe_gemm_split.h)
#include <string.h>
#include "txe_common.h"
#include "txe_gemm.h"
#include "txe_gemm_split.h"

namespace txe {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16s;

// ---- packing: fp32 [rows][ld] -> three bf16 planes in fragment order --------------------------------------------------------------
// one thread per (fragment block rb, k-tile kt, lane): reads 8 consecutive floats of its slot's row, writes 16 bytes per plane
// A first GATLayer's input NEVER WRITTEN: element (row, c) of X = dropout([h | Emb[pos]]) formed where it is packed --
//   c < Kh: h[row][c];  Kh <= c < Kh + Pd: P[pos[row]][c - Kh];  beyond: 0;  times keep(mask word (row, c / 32), bit c % 32) * scale
// -- the arithmetic of build_x_job (txe_project.hip), so the planes are those of the stored X bit for bit.  h == NULL: the matrix at src.
// (struct SplitVSrc: txe_gemm_split.h)
// NC consecutive columns c0 .. c0 + NC - 1 (c0 a multiple of NC, NC | 32: one mask word per row) of R rows.  The kind of column range
// is decided ONCE, outside the row loops, and every load of a kind is issued before the first use (a branch between a load and its
// use costs a full round trip per row: csrc/txe_gather.h).
template <int NC, int R>
__device__ __forceinline__ void vsrc_load_rows(const SplitVSrc& v, const int (&row)[R], const int c0, float (&x)[R][NC]) {
    unsigned word[R];
    const bool masked = v.mask != nullptr && c0 < v.wpr * 32;
#pragma unroll
    for (int r = 0; r < R; ++r) word[r] = masked ? v.mask[(long long)row[r] * v.wpr + (c0 >> 5)] : 0xFFFFFFFFu;
    if (c0 + NC <= v.Kh) {                                           // feature columns
        const bool two = (NC & 1) == 0 && ((reinterpret_cast<uintptr_t>(v.h + c0) & 7) == 0) && (v.ld_h & 1) == 0;
        if (two) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float* hr = v.h + (long long)row[r] * v.ld_h + c0;
#pragma unroll
                for (int q = 0; q + 1 < NC; q += 2) { const float2 t = *reinterpret_cast<const float2*>(hr + q); x[r][q] = t.x; x[r][q + 1] = t.y; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int q = 0; q < NC; ++q) x[r][q] = v.h[(long long)row[r] * v.ld_h + c0 + q];
        }
    } else if (c0 >= v.Kh + v.Pd) {                                  // zero padding columns
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < NC; ++q) x[r][q] = 0.f;
        return;
    } else {                                                         // position columns, or the range that straddles Kh
        int pr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) pr[r] = (v.Pd > 0) ? v.pos[row[r]] : 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = c0 + q;
                const float hv = v.h[(long long)row[r] * v.ld_h + min(c, v.Kh - 1)];
                const float pv = (v.Pd > 0) ? v.P[(long long)pr[r] * v.Pd + min(max(c - v.Kh, 0), v.Pd - 1)] : 0.f;
                x[r][q] = (c < v.Kh) ? hv : ((c < v.Kh + v.Pd) ? pv : 0.f);
            }
    }
    if (masked) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < NC; ++q) x[r][q] = ((word[r] >> ((c0 & 31) + q)) & 1u) ? x[r][q] * v.scale : 0.f;
    }
}
struct SplitPackArgs { const float* src; long long ld; int rows, cols, side, nrb, nkt; uint4* dst; int transposed; SplitVSrc v; };
__device__ __forceinline__ void split_pack_job(const SplitPackArgs& a, const int bid, const int nb) {
    const float* __restrict__ src = a.src;
    uint4* __restrict__ dst = a.dst;
    const int nkt = a.nkt, rows = a.rows, cols = a.cols;
    const long long ld = a.ld;
    const long long total = (long long)a.nrb * nkt * 64;
    for (long long i = (long long)bid * blockDim.x + threadIdx.x; i < total; i += (long long)nb * blockDim.x) {
        const int l = (int)(i & 63);
        const long long f = i >> 6;                       // fragment pair index rb * nkt + kt
        const int kt = (int)(f % nkt), rb = (int)(f / nkt);
        const int row = split_slot_row(a.side, rb, l & 31), k0 = kt * SPL_KT + 8 * (l >> 5);
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = 0.f;
        if (a.v.h != nullptr) {
            if (row < rows) {
                const int r1[1] = {row};
                float x1[1][8];
                vsrc_load_rows<8, 1>(a.v, r1, k0, x1);
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = x1[0][q];
            }
        } else if (a.transposed) {                         // element (row, k) = src[k][row]: an operand given as its transpose
            if (row < rows) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k0 + q < cols) x[q] = src[(long long)(k0 + q) * ld + row];
            }
        } else if (row < rows) {
            const float* p = src + (long long)row * ld + k0;
            if (k0 + 7 < cols && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                const float4 u = *reinterpret_cast<const float4*>(p), v = *reinterpret_cast<const float4*>(p + 4);
                x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k0 + q < cols) x[q] = p[q];
            }
        }
        uint4 w1, w2, w3;
        if (__ballot(split_exceptional8(x)) != 0ull) split_raw8(x, w1, w2, w3);      // (a wave = one fragment: txe_gemm_split.h "the whole fp32 domain")
        else split3x8(x, w1, w2, w3);
        uint4* o = dst + f * 3 * 64 + l;
        o[0] = w1; o[64] = w2; o[128] = w3;
    }
}
__global__ __launch_bounds__(256) void split_pack_kernel(const SplitPackArgs a) { split_pack_job(a, blockIdx.x, gridDim.x); }

// ---- the product -----------------------------------------------------------------------------------------------------------------
// A (64 MI) x 128 tile per workgroup of four waves (2 x 2; a wave owns MI x 2 MFMA blocks of 32 x 32: 16 MI x 2 accumulator registers);
// k-tiles of 16 columns: (2 MI + 4) x 3 fragments per stage, NST stages in a ring, every wave copies a quarter of a stage's fragments
// global -> LDS directly per k-tile, reads 3 (MI + 2) with ds_read_b128 and issues 12 MI MFMAs.
struct SplitGemm {
    const char* A; const char* B;     // packed operands
    int nkt;                          // k-tiles of 16
    float* C; long long ldc; int M, N;
    int nbm, nbn, apply_exp;
    // epilogue extras (EPI kernels; txe_gemm.h epi_store_one): C = acc * (keep bit ? drop_scale : 0) * (act_src > 0 || column >= cols_act ? 1 : slope)
    const unsigned* mask; int mask_ld, mask_col0, mask_on; float drop_scale;
    const float* act_src; long long ld_act; float act_slope; int act_on, cols_act;
};

// A tile that saw an exceptional fragment (or whose result is not finite) recomputes itself: C tile [64 MI][128] into LDS (the stage ring
// is free after the k-loop), every output an fp32 FMA chain in ascending k over operands decoded from the packed form -- exact fp32
// values, IEEE arithmetic (txe_gemm_split.h "the whole fp32 domain").  Scalar speed and deliberately small code: thread t owns column
// t & 127 and every second row; it re-reads both operands' lane words from global memory (L2) for every output and half k-tile.
template <int MI>
__device__ __forceinline__ void gemm_nt_split_slow_tile(const char* __restrict__ A, const char* __restrict__ B, const int nkt, const int tm,
                                                        const int tn, float* __restrict__ Cs) {
    constexpr int NA = 2 * MI, NB = 4;
    const int c = threadIdx.x & 127;
    const uint4* fb0 = reinterpret_cast<const uint4*>(B + ((long long)(NB * tn + 2 * (c >> 6) + (c & 1)) * nkt * 3) * SPL_FRAG_BYTES) + ((c & 63) >> 1);
#pragma unroll 1
    for (int r = threadIdx.x >> 7; r < 64 * MI; r += 2) {
        const uint4* fa0 = reinterpret_cast<const uint4*>(A + ((long long)(NA * tm + (r >> 5)) * nkt * 3) * SPL_FRAG_BYTES) + (r & 31);
        float sum = 0.f;
#pragma unroll 1
        for (int h = 0; h < 2 * nkt; ++h) {              // half k-tiles: fragment triple h >> 1, lanes 32 (h & 1) + slot
            const uint4* fa = fa0 + (h >> 1) * 192 + (h & 1) * 32;
            const uint4* fb = fb0 + (h >> 1) * 192 + (h & 1) * 32;
            float a[8], b[8];
            split_decode8(fa[0], fa[64], fa[128], a);
            split_decode8(fb[0], fb[64], fb[128], b);
#pragma unroll
            for (int q = 0; q < 8; ++q) sum = __builtin_fmaf(a[q], b[q], sum);
        }
        Cs[r * 129 + c] = sum;
    }
}

// EPI: 0 = plain stores from the accumulators; 1 = the same with the dropout-mask / activation factors; 2 = the tile goes through LDS into
// gemm_kernel's own epilogue (txe_gemm.h gemm_tile_epilogue: exp, pick, count and best-k modes -- the scoring loop), E = its arguments
template <int MI, int NST, int MINB, int EPI = 0>
__global__ __launch_bounds__(256, MINB) void gemm_nt_split_kernel(const SplitGemm p, const Epi E) {
    constexpr int NA = 2 * MI, NB = 4;                      // A / B fragment blocks per tile
    constexpr int NF = 3 * (NA + NB), CP = (NF + 3) / 4;    // fragments per stage, copies per wave and k-tile (the last wave: the rest)
    static_assert(NST == 2 || NF % 4 == 0, "the three-stage ring waits on an exact copy count");
    constexpr int STAGE_U4 = NF * 64;
    __shared__ __attribute__((aligned(16))) uint4 smem_u4[NST * STAGE_U4];
    uint4* const st0 = smem_u4;
    uint4* const st1 = smem_u4 + STAGE_U4;
    uint4* const st2 = smem_u4 + (NST == 3 ? 2 : 0) * STAGE_U4;
    static_assert(EPI != 2 || (MI == 2 && NST * STAGE_U4 * 4 >= 128 * 132 + 128 * 10), "the epilogue's C tile + count scratch live in the stages");
    typedef __attribute__((address_space(3))) uint4 lds_u4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    // tile order: blocks of 16 row panels x 8 column tiles, column-tile fast -- the 64 tiles an XCD runs at a time share HALF of a
    // 16-tile-wide B (2 MB of planes at K = 320) and 8 A panels (2 MB): its 4 MB L2 holds them, where 4 panels x all 16 column tiles
    // (row-major order) cycled 4.9 MB through it (FETCH_SIZE 304 MB per launch for 38 MB of operands)
    int tm, tn;
    {
        constexpr int GM = 16, GN = 8;
        const int srt = GM * p.nbn, sr = lb / srt, rem = lb - sr * srt;
        const int hgt = min(GM, p.nbm - sr * GM);
        const int cg = rem / (hgt * GN), r = rem - cg * hgt * GN, wg = min(GN, p.nbn - cg * GN);
        tm = sr * GM + r / wg; tn = cg * GN + r % wg;
    }
    const int wm = w >> 1, wn = w & 1;
    const int nkt = p.nkt;
    // (EPI 2) count / pick modes: as gemm_kernel -- a pick-mode tile without a (row, own column) pair has nothing to do; the rows'
    // positive ranges and first thresholds are fetched now, under the k-loop
    int cnt_pb = 0, cnt_np = 0;
    float cnt_th[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == 2) {
        const int m0 = tm * 128, n0 = tn * SPL_BN;
        if (E.cnt_mode == 3) {
            const int lo = E.cnt_off[m0], hi = E.cnt_off[min(m0 + 128, p.M)];
            if (n0 >= hi || n0 + SPL_BN <= lo) return;
        }
        if ((E.cnt_mode == 1 || E.cnt_mode == 2) && threadIdx.x < 128 && m0 + (int)threadIdx.x < p.M) {
            cnt_pb = E.cnt_off[m0 + threadIdx.x];
            cnt_np = E.cnt_off[m0 + threadIdx.x + 1] - cnt_pb;
#pragma unroll
            for (int k = 0; k < 8; ++k) cnt_th[k] = E.cnt_thr[(k < cnt_np) ? cnt_pb + k : 0];
        }
    }
    // this wave's copies: fragments f = CP w + q of the stage; f < 3 NA: A block f / 3, plane f % 3, else B block (f - 3 NA) / 3.
    // Source of k-tile kt: fragment (block, kt, plane) of the packed operand (wave-uniform base + lane * 16)
    const char* gsrc[CP];
#pragma unroll
    for (int q = 0; q < CP; ++q) {
        const int f = min(CP * w + q, NF - 1);
        const bool isa = f < 3 * NA;
        const int blk = isa ? f / 3 : (f - 3 * NA) / 3, pl = isa ? f % 3 : (f - 3 * NA) % 3;
        const char* base = isa ? p.A : p.B;
        const long long b0 = isa ? (long long)NA * tm + blk : (long long)NB * tn + blk;
        gsrc[q] = base + (b0 * nkt) * (3 * SPL_FRAG_BYTES) + pl * SPL_FRAG_BYTES;
    }
    const unsigned lane_off = l * 16;

    f32x16s acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

// (inline asm: hipcc's wait-count pass treats every pending LDS-direct copy as a hazard for every later ds_read and turns the ring into
//  vmcnt(0) -- copy, wait, compute -- however the stages are declared; issued from asm the copies are invisible to it and the waits
//  below are the only ones; nothing else in this kernel uses M0 -- gfx9 DS instructions do not)
#define TXE_SP_COPY(g_, d_)                                                                                           \
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(g_), "s"((unsigned)(uintptr_t)(d_)) : "memory");
#define TXE_SP_ISSUE(st_, kt_)                                                                                        \
    {                                                                                                                \
        const long long ko = (long long)min((kt_), nkt - 1) * (3 * SPL_FRAG_BYTES);                                  \
        lds_u4* d0 = (lds_u4*)(st_) + (CP * w) * 64;                                                                 \
        _Pragma("unroll") for (int q = 0; q < CP; ++q)                                                               \
            if (NF % 4 == 0 || CP * w + q < NF) TXE_SP_COPY(gsrc[q] + ko, d0 + q * 64)                               \
    }
#define TXE_SP_MFMA(pa_, pb_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa_], fb[j][pb_], acc[i][j], 0, 0, 0);
#define TXE_SP_PRODUCTS()                                                                                             \
    TXE_SP_MFMA(2, 0) TXE_SP_MFMA(0, 2) TXE_SP_MFMA(1, 1) TXE_SP_MFMA(1, 0) TXE_SP_MFMA(0, 1) TXE_SP_MFMA(0, 0)
#define TXE_SP_COMPUTE(st_)                                                                                           \
    {                                                                                                                \
        bf16x8 fa[MI][3], fb[2][3];                                                                                  \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                               \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                          \
                const uint4 t = (st_)[((MI * wm + i) * 3 + q) * 64 + l];                                             \
                fa[i][q] = __builtin_bit_cast(bf16x8, t);                                                            \
            }                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                          \
                const uint4 t = (st_)[(3 * NA + (2 * wn + j) * 3 + q) * 64 + l];                                     \
                fb[j][q] = __builtin_bit_cast(bf16x8, t);                                                            \
            }                                                                                                        \
        TXE_SP_PRODUCTS()                                                                                            \
    }
    // this wave's copies of the stage about to be read have landed (three stages: the CP of the stage after it may still be in flight);
    // the barrier then says the same of every wave's, and that every wave is done reading the stage the next copies overwrite
#define TXE_SP_SYNC()                                                                                                 \
    {                                                                                                                \
        if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CP) : "memory");                            \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        __syncthreads();                                                                                             \
    }
#define TXE_SP_STEP(cur_, nxt_, kt_)                                                                                  \
    TXE_SP_SYNC()                                                                                                    \
    TXE_SP_ISSUE(nxt_, kt_)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    TXE_SP_COMPUTE(cur_)                                                                                             \
    __builtin_amdgcn_sched_barrier(0);

    TXE_SP_ISSUE(st0, 0)
    if constexpr (NST == 3) {
        TXE_SP_ISSUE(st1, 1)
        for (int t = 0; t < nkt; t += 3) {
            TXE_SP_STEP(st0, st2, t + 2)
            if (t + 1 >= nkt) break;
            TXE_SP_STEP(st1, st0, t + 3)
            if (t + 2 >= nkt) break;
            TXE_SP_STEP(st2, st1, t + 4)
        }
    } else {
        for (int t = 0; t < nkt; t += 2) {
            TXE_SP_STEP(st0, st1, t + 1)
            if (t + 1 >= nkt) break;
            TXE_SP_STEP(st1, st0, t + 2)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped copies past the last k-tile have landed too
#undef TXE_SP_STEP
#undef TXE_SP_SYNC
#undef TXE_SP_COMPUTE
#undef TXE_SP_MFMA
#undef TXE_SP_PRODUCTS
#undef TXE_SP_ISSUE
#undef TXE_SP_COPY

    {   // an accumulator that is not finite: a raw fragment's NaN plane, or an overflowing result -- the tile recomputes itself in fp32
        bool odd = false;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) odd |= not_finite(acc[i][j][e]);
        static_assert(NST * STAGE_U4 * 4 >= 64 * MI * 129, "the recomputed tile lives in the stage ring");
        if (__syncthreads_or(odd)) {                     // (the barrier: every wave is done reading the stages)
            float* Cs = reinterpret_cast<float*>(smem_u4);
            gemm_nt_split_slow_tile<MI>(p.A, p.B, nkt, tm, tn, Cs);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        acc[i][j][e] = Cs[(32 * MI * wm + 32 * i + 4 * (l >> 5) + (e & 3) + 8 * (e >> 2)) * 129 + 64 * wn + 2 * (l & 31) + j];
            __syncthreads();
        }
    }

    if constexpr (EPI == 2) {
        float* Cs = reinterpret_cast<float*>(smem_u4);
        __syncthreads();                                 // every wave is done reading the stages
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = 32 * MI * wm + 32 * i + 4 * (l >> 5) + (e & 3) + 8 * (e >> 2);
                    Cs[row * 132 + 64 * wn + 2 * (l & 31) + j] = acc[i][j][e];
                }
        __syncthreads();
        gemm_tile_epilogue<128, NST * STAGE_U4 * 4>(E, Cs, tm * 128, tn * SPL_BN, p.M, p.N, tn, p.nbn, 0, cnt_pb, cnt_np, cnt_th);
        return;
    }
    // accumulator register e of block (i, j): row 32 i + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), slot lane & 31 of B fragment 2 wn + j =
    // column 64 wn + 2 (lane & 31) + j of the tile
    const int c0 = tn * SPL_BN + 64 * wn + 2 * (l & 31);
    const int r0 = tm * (64 * MI) + 32 * MI * wm + 4 * (l >> 5);
    const bool vec = ((p.ldc & 1) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 7) == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = r0 + 32 * i + (e & 3) + 8 * (e >> 2);
            if (m < p.M) {
                float* dst = p.C + (long long)m * p.ldc + c0;
                float v0 = acc[i][0][e], v1 = acc[i][1][e];
                if constexpr (EPI == 1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int c = c0 + q;
                        if (c >= p.N) continue;
                        float g = 1.f;
                        if (p.mask_on) {
                            const int cm = c + p.mask_col0;
                            g = ((p.mask[(long long)m * p.mask_ld + (cm >> 5)] >> (cm & 31)) & 1u) ? p.drop_scale : 0.f;
                        }
                        if (p.act_on && c < p.cols_act && !(p.act_src[(long long)m * p.ld_act + c] > 0.f)) g *= p.act_slope;
                        if (q == 0) v0 *= g; else v1 *= g;
                    }
                }
                if (p.apply_exp) { v0 = __expf(v0); v1 = __expf(v1); }       // (the scoring loop's exp: the very call gemm_tile_epilogue makes)
                if (vec && c0 + 1 < p.N) *reinterpret_cast<float2*>(dst) = make_float2(v0, v1);
                else {
                    if (c0 < p.N) dst[0] = v0;
                    if (c0 + 1 < p.N) dst[1] = v1;
                }
            }
        }
}

// ---- side 2 packing: fp32 [rows][ld] -> contraction-major planes (txe_gemm_split.h) ------------------------------------------------
// one thread per (row tile nt, column tile h, lane): 8 rows x (4 + 1) columns -> five fragments' lane words per plane
struct SplitPackTArgs { const float* src; long long ld; int rows, cols, nht, nnt; uint4* dst; SplitVSrc v; };
__device__ __forceinline__ void split_pack_t_job(const SplitPackTArgs& a, const int bid, const int nb) {
    const float* __restrict__ src = a.src;
    uint4* __restrict__ dst = a.dst;
    const int rows = a.rows, nht = a.nht;
    const long long ld = a.ld;
    const long long total = (long long)a.nnt * nht * 64;
    const int nkb = nht * 5;
    for (long long i = (long long)bid * blockDim.x + threadIdx.x; i < total; i += (long long)nb * blockDim.x) {
        const int l = (int)(i & 63), s = l & 31, nh = l >> 5;
        const long long g = i >> 6;
        const int h = (int)(g % nht), nt = (int)(g / nht);
        const int n0 = nt * 16 + nh * 8;
        float4 q[8];
        float o[8];
        const int c4 = min(160 * h + 4 * s, a.cols - 4), c1 = min(160 * h + 128 + s, a.cols - 1);   // (clamped; zeroed below)
        const bool ok4 = 160 * h + 4 * s + 3 < a.cols, ok1 = 160 * h + 128 + s < a.cols;
        if (a.v.h != nullptr) {
            int rr[8];
            float t4[8][4], t1[8][1];
#pragma unroll
            for (int r = 0; r < 8; ++r) rr[r] = min(n0 + r, rows - 1);
            vsrc_load_rows<4, 8>(a.v, rr, c4, t4);
            vsrc_load_rows<1, 8>(a.v, rr, c1, t1);
#pragma unroll
            for (int r = 0; r < 8; ++r) { q[r] = make_float4(t4[r][0], t4[r][1], t4[r][2], t4[r][3]); o[r] = t1[r][0]; }
        } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = min(n0 + r, rows - 1);
            const float* p = src + (long long)row * ld;
            q[r] = *reinterpret_cast<const float4*>(p + c4);
            o[r] = p[c1];
        }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (n0 + r >= rows || !ok4) q[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + r >= rows || !ok1) o[r] = 0.f;
        }
        uint4* base = dst + ((long long)nt * nkb + 5 * h) * 3 * 64 + l;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float x[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = j == 0 ? q[r].x : (j == 1 ? q[r].y : (j == 2 ? q[r].z : (j == 3 ? q[r].w : o[r])));
            uint4 w1, w2, w3;
            if (__ballot(split_exceptional8(x)) != 0ull) split_raw8(x, w1, w2, w3);  // (a wave = the five fragments of one (nt, h): each judged alone)
            else split3x8(x, w1, w2, w3);
            base[(j * 3 + 0) * 64] = w1; base[(j * 3 + 1) * 64] = w2; base[(j * 3 + 2) * 64] = w3;
        }
    }
}
__global__ __launch_bounds__(256) void split_pack_t_kernel(const SplitPackTArgs a) { split_pack_t_job(a, blockIdx.x, gridDim.x); }

// the packs a layer's two products need, in ONE launch: side 0 of X, side 1 of Wp, side 2 of X (three launches of 4-13 us cost their
// dispatch gaps on top)
struct SplitPackMulti { SplitPackArgs a[2]; SplitPackTArgs t; int nb[3]; };
__global__ __launch_bounds__(256) void split_pack_multi_kernel(const SplitPackMulti m) {
    int b = blockIdx.x;
    if (b < m.nb[0]) { split_pack_job(m.a[0], b, m.nb[0]); return; }
    b -= m.nb[0];
    if (b < m.nb[1]) { split_pack_job(m.a[1], b, m.nb[1]); return; }
    split_pack_t_job(m.t, b - m.nb[1], m.nb[2]);
}

// ---- the TN product ------------------------------------------------------------------------------------------------------------
// 128 (M) x 160 (N) tile per workgroup of four waves; wave w owns the 32 row slots of A block w x all five B blocks (80 accumulator
// registers).  k-tiles of 16 contraction rows, two fragment stages.  B's 15 fragments arrive by LDS-direct copies (15 KB contiguous,
// one k-tile ahead).  A's [16][128] fp32 patch arrives by LDS-direct copies too, TWO k-tiles ahead, into a raw ring: wave w copies rows
// 4 w .. 4 w + 3 (2 KB) and is the only reader of them -- thread (column pair q = lane, row quad w) reads its 4 rows x 2 adjacent
// columns back, splits them and writes six 8-byte half lane words into the next fragment stage while the other waves still multiply.
// (The first version loaded the patch into registers one k-tile ahead: the wait for those loads cost 25 of its 131 us.)
// Every copy is issued from inline asm and every wave issues exactly 6 per k-tile (4 B + 2 A), so the waits are exact counts.
// A's slot permutation: slot s of block fb is column 64 (fb >> 1) + 2 s + (fb & 1) of the tile (a thread's two columns land in slot
// s of two neighbouring blocks: consecutive lanes write consecutive LDS words).
struct SplitTn {
    const float* A; long long lda; const char* Bt; int nkb;
    float* C; long long ldc, split_stride;
    int n_rows, ksplit, ntm, ntn, N;
};
constexpr int SPT_A_U4 = 4 * 3 * 64, SPT_B_U4 = 5 * 3 * 64, SPT_STAGE_U4 = SPT_A_U4 + SPT_B_U4;   // 12 KB + 15 KB
constexpr int SPT_RAW_F = 16 * 128;                                                              // 8 KB

// The TN tile's recomputation (txe_gemm_split.h "the whole fp32 domain"): the 32 C rows of A block `blk` x 160 columns into LDS, every
// output an fp32 FMA chain over the slice's contraction rows in ascending order -- A read as the fp32 it is, B decoded from its packed form
__device__ __forceinline__ void gemm_tn_split_slow_block(const SplitTn& p, const int tm, const int h, const int kbeg, const int kend,
                                                         const int blk, float* __restrict__ Cs) {
#pragma unroll 1
    for (int o = threadIdx.x; o < 32 * 160; o += 256) {
        const int sa = o / 160, cc = o - sa * 160;
        const int j = cc < 128 ? (cc & 3) : 4, sb = cc < 128 ? (cc >> 2) : cc - 128;
        const float* a = p.A + (long long)tm * 128 + 64 * (blk >> 1) + 2 * sa + (blk & 1);
        float sum = 0.f;
#pragma unroll 1
        for (int n0 = kbeg; n0 < kend; n0 += 8) {        // (kbeg is a multiple of 16: eight rows = one half of a fragment's lanes)
            const uint4* f = reinterpret_cast<const uint4*>(p.Bt + ((((long long)(n0 >> 4) * p.nkb + 5 * h + j) * 3) * SPL_FRAG_BYTES)) + ((n0 >> 3) & 1) * 32 + sb;
            float b[8];
            split_decode8(f[0], f[64], f[128], b);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (n0 + q < kend) sum = __builtin_fmaf(a[(long long)(n0 + q) * p.lda], b[q], sum);
        }
        Cs[sa * 161 + cc] = sum;
    }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_split_kernel(const SplitTn p) {
    __shared__ __attribute__((aligned(16))) uint4 st0[SPT_STAGE_U4];
    __shared__ __attribute__((aligned(16))) uint4 st1[SPT_STAGE_U4];
    __shared__ __attribute__((aligned(16))) float raw0[SPT_RAW_F];
    __shared__ __attribute__((aligned(16))) float raw1[SPT_RAW_F];
    typedef __attribute__((address_space(3))) uint4 lds_u4;
    typedef __attribute__((address_space(3))) float lds_f;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int ntiles = p.ntm * p.ntn;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int z = vb / ntiles, lb = vb % ntiles;
    const int tm = lb / p.ntn, h = lb % p.ntn;
    const int kbeg = z * p.ksplit, kend = min(p.n_rows, kbeg + p.ksplit);
    const int nk = kend > kbeg ? (kend - kbeg + 15) / 16 : 0;
    // B: fragments 4 w .. 4 w + 3 of the k-tile's 15 (wave 3: 12, 13, 14 and 14 again -- every wave issues the same number of copies)
    const char* gb = p.Bt + (((long long)(kbeg / 16) * p.nkb + 5 * h) * 3 + 4 * w) * SPL_FRAG_BYTES;
    const long long adv_b = (long long)p.nkb * 3 * SPL_FRAG_BYTES;
    const unsigned lane_off = l * 16;
    // A: rows 4 w + 2 i + (lane >> 5), i = 0, 1, 16 bytes per lane (32-bit byte offsets from the tile's first column: the launcher checks)
    const float* ga = p.A + (long long)tm * 128;
    const unsigned a_col = (l & 31) * 16;
    const int last_row = p.n_rows - 1;
    const int g = l >> 5, s = l & 31;

    f32x16s acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#define TXE_ST_COPY(v_, g_, d_)                                                                                       \
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(v_), "s"(g_), "s"((unsigned)(uintptr_t)(d_)) : "memory");
#define TXE_ST_ISSUE_B(st_, t_)                                                                                       \
    {                                                                                                                \
        const char* sb = gb + (long long)min((t_), nk - 1) * adv_b;                                                  \
        lds_u4* d0 = (lds_u4*)(st_) + SPT_A_U4 + (4 * w) * 64;                                                       \
        const int q3 = w < 3 ? 3 : 2;                                                                                \
        TXE_ST_COPY(lane_off, sb, d0)                                                                                \
        TXE_ST_COPY(lane_off, sb + SPL_FRAG_BYTES, d0 + 64)                                                          \
        TXE_ST_COPY(lane_off, sb + 2 * SPL_FRAG_BYTES, d0 + 128)                                                     \
        TXE_ST_COPY(lane_off, sb + q3 * SPL_FRAG_BYTES, d0 + q3 * 64)                                                \
    }
#define TXE_ST_ISSUE_A(raw_, t_)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
        const int row = min(kbeg + min((t_), nk - 1) * 16 + 4 * w + 2 * i + (l >> 5), last_row);                     \
        const unsigned off = (unsigned)row * (unsigned)(p.lda * 4) + a_col;                                          \
        TXE_ST_COPY(off, ga, (lds_f*)(raw_) + (4 * w + 2 * i) * 128)                                                 \
    }
    // raw rows 4 w .. 4 w + 3, columns 2 lane, 2 lane + 1 = half (w & 1) of lane word (nh = w >> 1, slot s) of blocks 2 g and 2 g + 1
#define TXE_ST_CONVERT(raw_, st_)                                                                                     \
    {                                                                                                                \
        float2 ra[4];                                                                                                \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) ra[r] = *reinterpret_cast<const float2*>((raw_) + (4 * w + r) * 128 + 2 * l); \
        const float x0[4] = {ra[0].x, ra[1].x, ra[2].x, ra[3].x}, x1[4] = {ra[0].y, ra[1].y, ra[2].y, ra[3].y};      \
        uint2 u1, u2, u3, v1, v2, v3;                                                                                \
        split3x4(x0, u1, u2, u3);                                                                                    \
        split3x4(x1, v1, v2, v3);                                                                                    \
        uint2* d = reinterpret_cast<uint2*>((st_) + (2 * g) * 3 * 64 + (w >> 1) * 32 + s) + (w & 1);                 \
        d[0] = u1; d[2 * 64] = u2; d[4 * 64] = u3;                                                                   \
        d[6 * 64] = v1; d[8 * 64] = v2; d[10 * 64] = v3;                                                             \
    }
#define TXE_ST_MFMA(pa_, pb_)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 5; ++j)                                                                    \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa_], fb[j][pb_], acc[j], 0, 0, 0);
#define TXE_ST_COMPUTE(st_)                                                                                           \
    {                                                                                                                \
        bf16x8 fa[3], fb[5][3];                                                                                      \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) fa[c] = __builtin_bit_cast(bf16x8, (st_)[(w * 3 + c) * 64 + l]); \
        _Pragma("unroll") for (int j = 0; j < 5; ++j)                                                                \
            _Pragma("unroll") for (int c = 0; c < 3; ++c)                                                            \
                fb[j][c] = __builtin_bit_cast(bf16x8, (st_)[SPT_A_U4 + (j * 3 + c) * 64 + l]);                       \
        TXE_ST_MFMA(2, 0) TXE_ST_MFMA(0, 2) TXE_ST_MFMA(1, 1) TXE_ST_MFMA(1, 0) TXE_ST_MFMA(0, 1) TXE_ST_MFMA(0, 0)  \
    }
    // step t: in flight at its start are B(t) (4 copies) and, younger, A(t+1) (2 copies)
#define TXE_ST_STEP(cur_, nxt_, rcur_, rnxt_, t_)                                                                     \
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     /* B(t) has landed */                                       \
    __syncthreads();                                     /* ... every wave's, and every wave's converted A(t) */     \
    TXE_ST_ISSUE_B(nxt_, (t_) + 1)                                                                                   \
    TXE_ST_ISSUE_A(rcur_, (t_) + 2)                      /* (this wave converted raw(t) a step ago) */               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    TXE_ST_COMPUTE(cur_)                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     /* A(t+1) has landed (this wave's own rows) */              \
    if ((t_) + 1 < nk) TXE_ST_CONVERT(rnxt_, nxt_)

    if (nk > 0) {
        TXE_ST_ISSUE_B(st0, 0)
        TXE_ST_ISSUE_A(raw0, 0)
        TXE_ST_ISSUE_A(raw1, 1)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        TXE_ST_CONVERT(raw0, st0)
        for (int t = 0; t < nk; t += 2) {
            TXE_ST_STEP(st0, st1, raw0, raw1, t)
            if (t + 1 >= nk) break;
            TXE_ST_STEP(st1, st0, raw1, raw0, t + 1)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped copies past the last k-tile
    }
#undef TXE_ST_STEP
#undef TXE_ST_COMPUTE
#undef TXE_ST_MFMA
#undef TXE_ST_CONVERT
#undef TXE_ST_ISSUE_A
#undef TXE_ST_ISSUE_B
#undef TXE_ST_COPY

    {   // exceptional operands (txe_gemm_split.h "the whole fp32 domain"): an accumulator that is not finite -- a raw B fragment's NaN plane;
        // Inf / NaN / beyond-bf16 elements of A; an overflowing result -- and the tile recomputes itself in fp32
        bool odd = false;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) odd |= not_finite(acc[j][e]);
        static_assert(SPT_STAGE_U4 * 4 >= 32 * 161, "a block's recomputed rows live in one fragment stage");
        if (__syncthreads_or(odd)) {
            float* Cs = reinterpret_cast<float*>(st0);
#pragma unroll 1
            for (int blk = 0; blk < 4; ++blk) {
                gemm_tn_split_slow_block(p, tm, h, kbeg, kend, blk, Cs);
                __syncthreads();
                if (blk == w) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float* row = Cs + ((e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 161;
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j][e] = row[4 * (l & 31) + j];
                        acc[4][e] = row[128 + (l & 31)];
                    }
                }
                __syncthreads();
            }
        }
    }

    // accumulator register e: row slot (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of A block w = C row 64 (w >> 1) + 2 slot + (w & 1) of the
    // tile; blocks 0-3: columns 4 (lane & 31) + j, block 4: column 128 + (lane & 31)
    float* cb = p.C + (long long)z * p.split_stride + (long long)(tm * 128 + 64 * (w >> 1) + (w & 1)) * p.ldc + 160 * h;
    const int sl = l & 31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int slot = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        float* dst = cb + (long long)(2 * slot) * p.ldc;
        if (160 * h + 4 * sl + 3 < p.N) *reinterpret_cast<float4*>(dst + 4 * sl) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
        if (160 * h + 128 + sl < p.N) dst[128 + sl] = acc[4][e];
    }
}

static bool fill_pack(SplitPackArgs& a, int& nb, const float* src, long long ld, int rows, int cols, int side, void* packed) {
    const int tr = side >> 1;                             // sides 2, 3 = sides 0, 1 of a matrix given as its transpose [cols][ld >= rows]
    if (!src || !packed || rows < 1 || cols < 1 || ld < (tr ? rows : cols) || side < 0 || side > 3) return false;
    a.src = src; a.ld = ld; a.rows = rows; a.cols = cols; a.side = side & 1; a.dst = (uint4*)packed; a.transposed = tr;
    memset(&a.v, 0, sizeof(a.v));                         // (a stored matrix; split_pack_layer_launch sets the sources of an unstored X)
    a.nrb = ((rows + 767) / 768) * 24; a.nkt = (cols + SPL_KT - 1) / SPL_KT;
    const long long total = (long long)a.nrb * a.nkt * 64;
    nb = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    return true;
}
static bool fill_pack_t(SplitPackTArgs& a, int& nb, const float* src, long long ld, int rows, int cols, void* packed) {
    if (!src || !packed || rows < 1 || cols < 4 || cols % 4 != 0 || ld < cols || (ld & 3) != 0 || (reinterpret_cast<uintptr_t>(src) & 15) != 0)
        return false;
    a.src = src; a.ld = ld; a.rows = rows; a.cols = cols; a.nht = (cols + 159) / 160; a.nnt = (rows + 15) / 16; a.dst = (uint4*)packed;
    memset(&a.v, 0, sizeof(a.v));
    const long long total = (long long)a.nnt * a.nht * 64;
    nb = (int)((total + 255) / 256);
    return true;
}
int split_pack_launch(const float* src, long long ld, int rows, int cols, int side, void* packed, hipStream_t stream) {
    SplitPackArgs a;
    int nb;
    if (!fill_pack(a, nb, src, ld, rows, cols, side, packed)) return TXE_ERR_ARG;
    ProfScope prof("split_pack_kernel", stream, 10.0 * rows * (double)cols, 1);
    hipLaunchKernelGGL(split_pack_kernel, dim3(nb), dim3(256), 0, stream, a);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}
// X [n][ldx] -> Xs (side 0) and, Xt != NULL, Xt (side 2); W [f][ldw] -> Ws (side 1): one launch
// (K columns for the NT operands, Kt_cols -- a multiple of 160 -- for the contraction-major one)
int split_pack_layer_launch(const float* X, long long ldx, int n, const float* W, long long ldw, int f, int K, int Kt_cols, void* Xs, void* Ws,
                            void* Xt, hipStream_t stream, const SplitVSrc* vs) {
    SplitPackMulti m;
    memset(&m, 0, sizeof(m));
    // (vs: X is not stored -- its elements are formed from vs; the fill functions still want a readable 16-byte aligned address)
    if (vs && (!vs->h || vs->Kh < 1 || vs->Pd < 0 || vs->ld_h < vs->Kh || (vs->Pd > 0 && (!vs->pos || !vs->P)) || (vs->mask && vs->wpr < 1))) return TXE_ERR_ARG;
    if (vs) X = reinterpret_cast<const float*>(W);
    if (!fill_pack(m.a[0], m.nb[0], X, ldx, n, K, 0, Xs) || !fill_pack(m.a[1], m.nb[1], W, ldw, f, K, 1, Ws)) return TXE_ERR_ARG;
    if (Xt && !fill_pack_t(m.t, m.nb[2], X, ldx, n, Kt_cols, Xt)) return TXE_ERR_ARG;
    if (vs) { m.a[0].v = *vs; m.t.v = *vs; }
    ProfScope prof("split_pack_multi_kernel", stream, 10.0 * K * ((Xt ? 2.0 : 1.0) * n + f), 1);
    hipLaunchKernelGGL(split_pack_multi_kernel, dim3(m.nb[0] + m.nb[1] + m.nb[2]), dim3(256), 0, stream, m);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int gemm_nt_split_launch(const void* Ap, const void* Bp, int M, int N, int K, float* C, long long ldc, double alg_flops, hipStream_t stream,
                         const SplitEpi* epi) {
    if (!Ap || !Bp || !C || M < 1 || N < 1 || K < 1 || ldc < N) return TXE_ERR_ARG;
    SplitGemm p;
    memset(&p, 0, sizeof(p));
    p.A = (const char*)Ap; p.B = (const char*)Bp; p.nkt = (K + SPL_KT - 1) / SPL_KT;
    p.C = C; p.ldc = ldc; p.M = M; p.N = N;
    p.nbm = (M + 127) / 128; p.nbn = (N + SPL_BN - 1) / SPL_BN;
    const Epi E0 = epi_plain(C, ldc, N);                // (unused by the EPI 0 / 1 instantiations)
    // (named as rocprofv3 prints the default instantiations: bench.py joins its HIP-event timings with the committed profiles by name)
    ProfScope prof(epi ? "gemm_nt_split_kernel<2, 3, 2, 1>" : "gemm_nt_split_kernel<2, 3, 2, 0>", stream, alg_flops > 0.0 ? alg_flops : 2.0 * M * (double)N * K, 0);
    const dim3 grid(p.nbm * p.nbn), blk(256);
    if (epi) {
        p.mask = epi->mask; p.mask_ld = epi->mask_ld; p.mask_col0 = epi->mask_col0; p.mask_on = epi->mask ? 1 : 0;
        p.drop_scale = epi->drop_scale;
        p.act_src = epi->act_src; p.ld_act = epi->ld_act; p.act_slope = epi->act_slope; p.act_on = epi->act_src ? 1 : 0; p.cols_act = epi->cols_act;
        hipLaunchKernelGGL((gemm_nt_split_kernel<2, 3, 2, 1>), grid, blk, 0, stream, p, E0);
        TXE_CHECK_LAUNCH();
        return TXE_OK;
    }
    hipLaunchKernelGGL((gemm_nt_split_kernel<2, 3, 2>), grid, blk, 0, stream, p, E0);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

// C = A B^T through gemm_kernel's epilogue E (E.c / E.ldc, exp, pick / count / best-k modes: the scoring loop's four entry points)
int gemm_nt_split_epi_launch(const void* Ap, const void* Bp, const Epi& E, int M, int N, int K, hipStream_t stream) {
    if (!Ap || !Bp || M < 1 || N < 1 || K < 1 || E.mask_on || E.act_on || E.c2) return TXE_ERR_ARG;
    SplitGemm p;
    memset(&p, 0, sizeof(p));
    p.A = (const char*)Ap; p.B = (const char*)Bp; p.nkt = (K + SPL_KT - 1) / SPL_KT;
    p.C = E.c; p.ldc = E.ldc; p.M = M; p.N = N;
    p.nbm = (M + 127) / 128; p.nbn = (N + SPL_BN - 1) / SPL_BN;
    Epi E2 = E;
    {   // the epilogue loads its extras unconditionally: give the unused ones a readable dummy address (gemm_launch_layout does the same)
        const void* valid = E2.c ? (const void*)E2.c : (const void*)Ap;
        E2.act_src = (const float*)valid;
        E2.mask = (const unsigned*)valid;
    }
    ProfScope prof("gemm_nt_split_kernel<2, 3, 2, 2>", stream, E.alg_flops > 0.0 ? E.alg_flops : 2.0 * M * (double)N * K, 0);
    // (plain / exp stores straight from the accumulators -- EPI 0 with the exp -- measured SLOWER than the LDS-staged 16-byte row stores:
    //  MAG-CS 200 against 215 G pairs/s)
    hipLaunchKernelGGL((gemm_nt_split_kernel<2, 3, 2, 2>), dim3(p.nbm * p.nbn), dim3(256), 0, stream, p, E2);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int split_pack_t_launch(const float* src, long long ld, int rows, int cols, void* packed, hipStream_t stream) {
    SplitPackTArgs a;
    int nb;
    if (!fill_pack_t(a, nb, src, ld, rows, cols, packed)) return TXE_ERR_ARG;
    ProfScope prof("split_pack_t_kernel", stream, 10.0 * rows * (double)cols, 1);
    hipLaunchKernelGGL(split_pack_t_kernel, dim3(nb), dim3(256), 0, stream, a);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int gemm_tn_split_launch(const float* A, long long lda, int M, const void* Bt, int N, int n_rows, int S, int ksplit, float* part, long long ldc,
                         long long split_stride, double alg_flops, hipStream_t stream) {
    if (!A || !Bt || !part || !split_tn_eligible(M, N) || n_rows < 1 || S < 1 || ksplit < 16 || ksplit % 16 != 0 || ldc < N || (ldc & 3) != 0 ||
        (lda & 3) != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0 || !split_tn_fits(n_rows, lda) || (reinterpret_cast<uintptr_t>(part) & 15) != 0 || (split_stride & 3) != 0)
        return TXE_ERR_ARG;
    SplitTn p;
    p.A = A; p.lda = lda; p.Bt = (const char*)Bt; p.nkb = ((N + 159) / 160) * 5; p.N = N;
    p.C = part; p.ldc = ldc; p.split_stride = split_stride;
    p.n_rows = n_rows; p.ksplit = ksplit; p.ntm = M / 128; p.ntn = (N + 159) / 160;
    ProfScope prof("gemm_tn_split_kernel", stream, alg_flops > 0.0 ? alg_flops : 2.0 * M * (double)N * n_rows, 0);
    hipLaunchKernelGGL(gemm_tn_split_kernel, dim3(p.ntm * p.ntn * S), dim3(256), 0, stream, p);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // namespace txe
using namespace txe;
extern "C" {

size_t txe_split_packed_bytes(int rows, int cols) { return (rows < 1 || cols < 1) ? 0 : split_packed_bytes(rows, cols); }

int txe_split_pack(const float* src, long long ld, int rows, int cols, int side, void* packed, void* stream) {
    return split_pack_launch(src, ld, rows, cols, side, packed, (hipStream_t)stream);
}

int txe_gemm_nt_split(const void* Ap, const void* Bp, int M, int N, int K, float* C, long long ldc, void* stream) {
    return gemm_nt_split_launch(Ap, Bp, M, N, K, C, ldc, 0.0, (hipStream_t)stream, nullptr);
}

size_t txe_split_packed_t_bytes(int rows, int cols) { return (rows < 1 || cols < 4) ? 0 : split_packed_t_bytes(rows, cols); }

int txe_split_pack_t(const float* src, long long ld, int rows, int cols, void* packed, void* stream) {
    return split_pack_t_launch(src, ld, rows, cols, packed, (hipStream_t)stream);
}

int txe_gemm_tn_split(const float* A, long long lda, int M, const void* Bt, int N, int n_rows, int S, int ksplit, float* part, long long ldc,
                      long long split_stride, void* stream) {
    return gemm_tn_split_launch(A, lda, M, Bt, N, n_rows, S, ksplit, part, ldc, split_stride, 0.0, (hipStream_t)stream);
}

}  // extern "C"
