// Z of the folded output layer straight from the projected feature TABLE (eval-mode encode) -- see gat_table_zsum_kernel
// in txe_gat.hip.
#pragma once
#include "txe_common.h"

namespace txe {

struct TabZsumArgs {
    const int* rowptr; const int* col;            // destination-sorted CSR of the batch
    const int* goff; int G;                       // graph offsets [G+1]
    const float* T; long long ld_t;               // table projection [n_table][ld_t] (features | a1 | a2 | pad)
    const int* rid; const int* pos; const float* T2; int vocab;     // table row / position row of every batch node; T2 [vocab][ld_t]
    int H, D; float attn_slope; int out_mode; float act_slope;      // the first layer's attention / the activation behind it
    const float* coef; const float* wsum;         // folded layer: c~ [N], S [G]
    const float* P; int Pd;                       // folded layer's position embedding [vocab][Pd] (NULL / 0: none)
    int Kp;                                       // row width of Z: H*D feature columns, Pd position columns, zero padding
    float* Z;                                     // [G][Kp]
};

int gat_table_zsum_supported(int H, int D, long long ld_t, int vocab, int Kp, int Pd);
int gat_table_zsum_launch(const TabZsumArgs& a, hipStream_t stream);

}  // namespace txe
