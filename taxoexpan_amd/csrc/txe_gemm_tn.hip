// Instantiates the fp32 MFMA GEMM kernels (txe_gemm.h) for the "tn" operand layout, and the LDS-direct split-K variant.
#include "txe_gemm.h"
#include "txe_gemm_tnlds.h"

namespace txe {

int gemm_tn(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s, void* tail_ws, size_t tail_ws_bytes) {
    return gemm_launch_layout<false, false>(A, B, E, M, N, K, splits, s, tail_ws, tail_ws_bytes);
}

int gemm_tn_lds_launch(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, int ksplit, hipStream_t stream) {
    auto plain = [](const VMat& m) {
        return m.p != nullptr && m.p2 == nullptr && m.p3 == nullptr && m.mask_on == 0 && m.cols_main == m.cols && m.rows_main >= m.rows &&
               (m.ld & 3) == 0 && m.ld >= 4 && (reinterpret_cast<uintptr_t>(m.p) & 15) == 0;
    };
    if ((E.route & GEMM_ROUTE_NO_TN_LDS) || !plain(A) || !plain(B) || E.act_on || E.mask_on || E.cnt_mode != 0 || E.apply_exp ||
        (E.c2 != nullptr && E.cols_main < N) || A.cols < M || B.cols < N || A.rows < K || B.rows < K || K < 1 || (E.ldc & 3) ||
        (reinterpret_cast<uintptr_t>(E.c) & 15) != 0 || (E.split_stride & 3))
        return TXE_ERR_ARG;
    TnLds p;
    p.A = A.p; p.lda = A.ld; p.M = M; p.B = B.p; p.ldb = B.ld; p.N = N; p.K = K; p.ksplit = ksplit;
    p.C = E.c; p.ldc = E.ldc; p.split_stride = E.split_stride;
    const int tiles = ((M + TL_BM - 1) / TL_BM) * ((N + TL_BN - 1) / TL_BN);
    ProfScope prof("gemm_tn_lds_kernel", stream, E.alg_flops > 0.0 ? E.alg_flops : 2.0 * M * (double)N * K, 0);
    hipLaunchKernelGGL(gemm_tn_lds_kernel, dim3(tiles, splits), dim3(256), 0, stream, p);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

}  // namespace txe
