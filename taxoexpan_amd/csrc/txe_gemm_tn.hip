// Instantiates the fp32 MFMA GEMM kernels (txe_gemm.h) for the "tn" operand layout.
#include "txe_gemm.h"

namespace txe {

int gemm_tn(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s, void* tail_ws, size_t tail_ws_bytes) {
    return gemm_launch_layout<false, false>(A, B, E, M, N, K, splits, s, tail_ws, tail_ws_bytes);
}

}  // namespace txe
