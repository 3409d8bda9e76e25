// Instantiates the fp32 MFMA GEMM kernels (txe_gemm.h) once for the three operand layouts.
#include "txe_gemm.h"

namespace txe {

int gemm_nt(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s) {
    return gemm_launch_layout<true, true>(A, B, E, M, N, K, splits, s);
}
int gemm_nn(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s) {
    return gemm_launch_layout<true, false>(A, B, E, M, N, K, splits, s);
}
int gemm_tn(const VMat& A, const VMat& B, const Epi& E, int M, int N, int K, int splits, hipStream_t s) {
    return gemm_launch_layout<false, false>(A, B, E, M, N, K, splits, s);
}

}  // namespace txe
