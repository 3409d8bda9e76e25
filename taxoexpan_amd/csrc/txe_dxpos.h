// First-layer d_X (position columns only) as ONE HBM stream over d_Y -- see txe_dxpos.hip.
#pragma once
#include "txe_common.h"

namespace txe {

constexpr int DXPOS_ROWS = 16;       // rows of d_Y per workgroup (= one 16x16x4 MFMA row block)
constexpr int DXPOS_MAXC = 64;       // widest column range [c0, c0 + NC) the kernel covers

struct DxPosArgs {
    const float* dY; long long ld_dy; int n_rows; int K;     // d_Y [n_rows][K], K a multiple of 128
    const float* Wp; long long ld_w; int Kp;                 // Wp [K][Kp]; columns [c0, c0 + NC) are used (c0 % 4 == 0, NC <= 64)
    int c0, NC;
    const unsigned* mask; long long mask_ld; int mask_on; float drop_scale;   // keep bits of the [n_rows][Kt] layer input (always readable)
    float* dX; long long ld_dx;                              // d_X [n_rows][Kp] (columns [c0, c0 + roundup(NC, 4)) written) or NULL
    const int* pos; int vocab, Pd, pcol0;                    // position classes; ppart[b][v][j] = sum_{rows of block b, pos == v} out[row][pcol0 + j]
    float* ppart;
};

static inline int dxpos_blocks(int n_rows) { return (n_rows + DXPOS_ROWS - 1) / DXPOS_ROWS; }
int dxpos_launch(const DxPosArgs& a, hipStream_t stream);

}  // namespace txe
