/* libtxe -- C ABI of the MI355X (gfx950) kernels behind TaxoExpan's propagation / readout / match path.
 *
 * The reference has no FFI of its own: the path sits behind Python classes in model/model_zoo.py that call DGL 0.4
 * and torch.  Each entry point below names the reference lines it replaces; taxoexpan_amd/model_zoo.py binds them
 * with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a BORROWED DEVICE pointer (fp32 / int32, contiguous unless a row stride `ld_*` is given, in
 *     elements); nothing is allocated, freed or synchronised inside; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*);  workspaces are supplied by the caller (size from the *_ws_bytes functions)
 *   - return value: 0 = TXE_OK, <0 = error (TXE_ERR_ARG -1, TXE_ERR_LAUNCH -2, TXE_ERR_WORKSPACE -3); never throws
 *   - re-entrant; no state between calls except the optional per-launch profiler (txe_profile_*, off by default: a process-global
 *     switch and a per-device ring of events) and the cached CU count of the current device
 *   - graph structure: destination-sorted CSR  (rowptr_in[N+1], col_src[E])  and source-sorted CSR
 *     (rowptr_out[N+1], col_dst[E], pos_out[E] = index of that edge in the destination-sorted order); per-edge
 *     arrays (alpha, dz) live in destination-sorted order
 *   - dropout: counter based splitmix64 hash (taxoexpan_amd/csrc/txe_common.h, restated for tests in
 *     taxoexpan_amd/rng.py): features use a precomputed keep-bit mask (txe_dropout_mask), attention coefficients
 *     hash (seed, csr_position*H+head) inline
 */
#ifndef TXE_H
#define TXE_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TXE_OK 0
#define TXE_ERR_ARG -1
#define TXE_ERR_LAUNCH -2
#define TXE_ERR_WORKSPACE -3

/* ---- feature dropout (nn.Dropout of model_zoo.py:36,82) as a keep-bit mask over an [n_rows][n_cols] operand: 32 columns per
 * word, ceil(n_cols/32) words per row.  Generated once per layer per step; forward, dX and dW all read the same mask.
 * Pass mask = NULL (or p = 0) to the projections for "no dropout" (eval mode). */
size_t txe_dropout_mask_bytes(long long n_rows, int n_cols);
int txe_dropout_mask(long long n_rows, int n_cols, float p, unsigned long long seed, unsigned* mask, void* stream);

/* ---- GATLayer dense part: model_zoo.py:82-85 (feat_drop, fc, a1, a2) with the PGAT concat of :214-215, on PADDED operands -------
 *   X  [N][Kp]  layer input [h (Kh) | Emb[pos] (Pd) | 0..], Kp = txe_gat_padded_k = roundup(Kh+Pd, 32).  txe_gat_build_x writes it
 *               (h == NULL: the feature part is already in place -- the previous layer's aggregation wrote it -- only the
 *               position-embedding and padding columns are filled).
 *   Wp [Fp][Kp] txe_gat_pack_weights: rows < H*D = fc.weight [H*D][Kh+Pd], rows H*D..H*D+2H = attention projections folded into
 *               the weights (a1 = h (W^T attn_l)), rest 0;  Fp = txe_gat_padded_f = roundup(H*D+2H, 128).
 *   Y  [N][Fp]  = dropout(X) Wp^T = [ft (H*D) | a1 (H) | a2 (H) | unused].
 * txe_gat_dense_bwd: d_Y [N][Fp] in the same layout with ZERO padding columns (txe_zero_cols) -> d_X columns [c0, Kh+Pd)
 * (c0 = 0 if need_dh, else the 4-aligned start of the position columns; columns < Kh x leaky'(X) if act_on -- the backward of
 * the F.leaky_relu of model_zoo.py:216 that produced h -- and all x the dropout factor), dW [H*D][Kh+Pd], d_attn_l/r [H*D],
 * dP [vocab][Pd].  ws: txe_gat_dense_ws_bytes (forward needs only txe_gemm_tail_ws_bytes, or NULL). */
int txe_gat_padded_k(int Kh, int Pd);
int txe_gat_padded_f(int H, int D);
int txe_gat_pack_weights(const float* W, const float* attn_l, const float* attn_r, int H, int D, int Kt, float* Wp, void* stream);
int txe_gat_build_x(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd, float* X, void* stream);

/* build_x + pack_weights + dropout_mask of one GATLayer (model_zoo.py:80-85) as ONE launch; same outputs as the three calls.
 * mask [n_nodes][ceil((Kh+Pd)/32)] may be NULL when feat_drop_p == 0. */
int txe_gat_layer_prepare(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd, float* X,
                          const float* W, const float* attn_l, const float* attn_r, int H, int D, float* Wp, float feat_drop_p,
                          unsigned long long seed, unsigned* mask, void* stream);

/* the same for every GATLayer of a stack in ONE launch: a layer's preparation depends on the parameters and on `pos` only, never on the
 * output of the layer below (whose aggregation writes the feature columns of X later), so the whole stack is prepared before its first
 * GEMM.  descs[i]: the arguments of txe_gat_layer_prepare for layer i (h == NULL for every layer but the first). */
struct txe_gat_prepare_desc {
    const float* h; long long ld_h; int n_nodes, Kh; const int* pos; const float* P; int Pd; float* X;
    const float *W, *attn_l, *attn_r; int H, D; float* Wp; float feat_drop_p; unsigned long long seed; unsigned* mask;
    int x_dropped;   /* 1: X is written with the feature dropout already applied -- the layer's GEMMs then take X as a plain operand:
                      * txe_gat_dense_fwd with feat_drop_p = 0 / mask = NULL, txe_gat_dense_bwd with x_dropped = 1 (the mask is still
                      * written: d_X's epilogue needs it).  With h == NULL the producer of the feature columns must drop them itself
                      * (txe_gat_aggregate_fwd with nx_mask and no nx_a12) */
};
int txe_gat_layers_prepare(const struct txe_gat_prepare_desc* descs, int n_layers, void* stream);

/* the same for a GCNLayer (model_zoo.py:35-37): txe_gat_build_x + txe_gcn_pack_weights + txe_dropout_mask as ONE launch */
int txe_gcn_layer_prepare(const float* h, long long ld_h, int n_nodes, int Kh, const int* pos, const float* P, int Pd, float* X,
                          const float* W, int Fo, float* Wp, float drop_p, unsigned long long seed, unsigned* mask, int x_dropped,
                          const float* bias_row /* NULL, or the layer's bias packed as row Kh + Pd of Wp (a padding row must exist) */,
                          void* stream);   /* x_dropped: as in txe_gat_prepare_desc (txe_gcn_dense_fwd then without mask) */
/* ... for every GCNLayer of a stack in ONE launch (descs[i]: the arguments of txe_gcn_layer_prepare for layer i; h == NULL for every layer
 * but the first), together with the stack's norm = in_degree^-1/2 (txe_gcn_norm, model_zoo.py:157-161; rowptr_in == NULL: without it) */
struct txe_gcn_prepare_desc {
    const float* h; long long ld_h; int n_nodes, Kh; const int* pos; const float* P; int Pd; float* X;
    const float* W; int Fo; float* Wp; float drop_p; unsigned long long seed; unsigned* mask; int x_dropped; const float* bias_row;
};
int txe_gcn_layers_prepare(const struct txe_gcn_prepare_desc* descs, int n_layers, const int* rowptr_in, int n_nodes, float* norm, void* stream);

/* Eval-mode layer-0 projection of a batch whose node features are rows of a taxonomy feature table (SURVEY 8f-2 "dedup by _id"):
 * the projection T = table W^T is formed once per DISTINCT taxonomy node (txe_gemm_plain), T2 = the position rows' projections, and
 * every batch node v gets Y[v] = T[row[v]] + T2[row2[v]].  n_cols % 4 == 0, 16-byte aligned rows; T2 / row2 may be NULL. */
int txe_gather_add_rows(const float* T, long long ld_t, const int* row, const float* T2, long long ld_t2, const int* row2, long long n_rows,
                        int n_cols, float* Y, long long ld_y, void* stream);
size_t txe_gat_dense_ws_bytes(int n_nodes, int Kh, int Pd, int H, int D, int vocab);
int txe_gat_dense_fwd(const float* X, int n_nodes, int Kh, int Pd, const float* Wp, int H, int D, float feat_drop_p,
                      const unsigned* mask, float* Y, void* ws, size_t ws_bytes, void* stream);
/* txe_gat_dense_fwd on the bf16 matrix pipe in fp32 accuracy (txe_gemm_nt_split below): X must be a PLAIN operand (dropout already
 * applied -- txe_gat_prepare_desc.x_dropped -- or none).  Xs / Ws = packed planes of X (side 0) / Wp (side 1), or NULL: they are then
 * packed into ws (txe_gat_dense_split_ws_bytes).  Xt_out (or NULL): txe_gat_dense_split_xt_bytes (0 = shape not eligible) for X packed
 * contraction-major (txe_split_pack_t) -- txe_gat_dense_bwd's `Xt`: its weight-gradient product then runs on the same pipe. */
size_t txe_gat_dense_split_ws_bytes(int n_nodes, int Kh, int Pd, int H, int D);
size_t txe_gat_dense_split_xt_bytes(int n_nodes, int Kh, int Pd, int H, int D);
int txe_gat_dense_fwd_split(const float* X, int n_nodes, int Kh, int Pd, const float* Wp, int H, int D, const void* Xs, const void* Ws,
                            void* Xt_out, float* Y, void* ws, size_t ws_bytes, void* stream);
/* ... for a FIRST layer whose input X = dropout([h | Emb[pos]]) (model_zoo.py:82,213-214) is never stored: the packs form its elements
 * from h [N][ld_h], the position table P [vocab][Pd] / pos [N] and the keep mask (NULL or feat_drop_p == 0: no dropout) -- the same
 * planes as packing the stored X, bit for bit.  txe_gat_layers_prepare with desc.X == NULL then writes mask and weights only, and
 * txe_gat_dense_bwd takes X == NULL with Xt (no act_on).  ws: txe_gat_dense_split_ws_bytes. */
int txe_gat_dense_fwd_split_src(const float* h, long long ld_h, const int* pos, const float* P, const unsigned* mask, float feat_drop_p,
                                int n_nodes, int Kh, int Pd, const float* Wp, int H, int D, void* Xt_out, float* Y, void* ws, size_t ws_bytes,
                                void* stream);
int txe_gat_dense_bwd(const float* X, int n_nodes, int Kh, int Pd, const int* pos, int vocab, const float* Wp, const float* W,
                      const float* attn_l, const float* attn_r, int H, int D, float feat_drop_p, const unsigned* mask, const float* d_Y,
                      int need_dh, int act_on, float act_slope, float* d_X, float* dW, float* d_attn_l, float* d_attn_r, float* dP,
                      int x_dropped, const void* Xt, int phases, void* chain, void* ws, size_t ws_bytes, void* stream);
/* phases | 16 and need_dh: d_X = d_Y Wp (the whole input gradient of a layer above the first) runs on the bf16 matrix pipe in fp32
 * accuracy, dropout mask and leaky' factor applied in its store loop; its packed operands live behind the workspace:
 * ws_bytes >= txe_gat_dense_ws_bytes + txe_gat_dense_bwd_split_ws_bytes, else TXE_ERR_WORKSPACE.  Without the bit: the fp32 MFMA,
 * whatever the size of the buffer -- the route is the caller's explicit choice. */
size_t txe_gat_dense_bwd_split_ws_bytes(int n_nodes, int Kh, int Pd, int H, int D);
/* phases: 7 = all of it; 1 | 2 | 4 = d_X | the weight-gradient product (split-K partial slices) | the reductions that finish dW,
 * d_attn, dP -- 1 and 2 are independent, 4 needs both.
 * chain (may be NULL): TXE_TAIL_CHAIN_BYTES of HOST memory, zero-filled = empty, owned by the caller for one backward pass.  The last
 * reduction launch of a layer ("phase B": dW / d_attn from the split-K slices, dP, d_pw) only finishes parameter gradients, so a
 * caller may DEFER it with phases | 64: the job is described in the chain instead of launched (its workspace and outputs must stay
 * alive), and the next call WITHOUT 64 that gets the chain -- the bottom layer's -- launches its own phase B and every deferred one
 * together.  txe_gat_tail_flush launches what a chain still holds. */
#define TXE_TAIL_CHAIN_BYTES 1024
int txe_gat_tail_flush(void* chain, void* stream);
int txe_zero_cols(float* x, long long ld, int n_rows, int c0, int c1, void* stream);
/* 1 when txe_gat_dense_bwd forms d_X with the streaming position-column kernel (a first PGAT layer: need_dh == 0, the columns behind
 * Kh fit 64 -- model_zoo.py:214-215): phase 1 is then ONE pass over d_Y at HBM speed that also leaves dP's per-class partial sums,
 * and belongs in line on the caller's stream (phases = 7), not beside the weight-gradient product on a second one. */
int txe_gat_dx_streams(int Kh, int Pd, int need_dh);

/* Eval-mode first GATLayer of a batch whose node features are rows of a feature table (test_fast.py:149-179 / infer.py:82-95
 * encode every taxonomy node many times): the projected rows ft[u] = T[rid[u]] + T2[pos[u]] (T = table x W^T [n_table][ld_t],
 * T2 = position embedding x W_p^T [vocab][ld_t], attention columns at H*D.. as in txe_gat_dense_fwd's output) are formed inside the
 * message/reduce sweep -- same arithmetic as txe_gather_add_rows followed by txe_gat_aggregate_fwd, without the [N][ld_t] round trip.
 * No dropout, alpha is not kept (inference).  nx_*: as in txe_gat_aggregate_fwd (no mask).  _supported: 1 if the shape fits (H <= 4,
 * 16-byte rows, the T2 rows and the folded rows in 56 KB of LDS), else the caller materialises the rows. */
int txe_gat_aggregate_table_supported(int H, int D, long long ld_t, int vocab, int nx_kp);
int txe_gat_aggregate_table_fwd(const int* rowptr_in, const int* col_src, int n_nodes, const float* T, long long ld_t, const int* rid,
                                const float* T2, const int* pos, int vocab, int H, int D, float attn_slope, int out_mode,
                                float act_slope, float* out, long long ld_out, const float* nx_wa, int nx_kp, float* nx_a12,
                                int npw, void* stream);
/* npw: as txe_gat_aggregate_fwd's (0 = chosen from the batch, 1 | 4 = one wave per node, 3 | 8..32 = the egonet walk, forced). */

/* ---- GATLayer message/reduce: model_zoo.py:90-95,106-114 (edge_attention, edge_softmax, attn_drop, update_all) -----
 * out_mode 0: out = aggregated features; 1: out = leaky_relu(aggregated, act_slope) (model_zoo.py:216 fused).
 * alpha [E][H] (post-softmax, pre-dropout; NULL in inference).
 * nx_a12 != NULL (optional fused epilogue, needs 16-byte aligned rows): `out` is the padded input X' [N][nx_kp] of the NEXT, one-head
 * GATLayer (ld_out == nx_kp, its position / padding columns already in place, nx_mask = its feature keep bits or NULL), and the
 * folded attention logits of that layer are formed on the way out: nx_a12[v][r] = <dropout(X'[v]), nx_wa[r]> (nx_wa [2][nx_kp] =
 * rows D, D+1 of its packed weights) -- txe_gat_collapse_fwd then runs with a12_ready = 1.
 * nx_a12 == NULL with nx_mask != NULL and nx_feat_drop_p > 0: `out` is the padded input of the NEXT GATLayer (ld_out == nx_kp, 16-byte
 * rows) and that layer's feature dropout is applied to the rows written (out = dropout(leaky_relu(aggregated))): its GEMMs then read a
 * plain operand (txe_gat_prepare_desc.x_dropped). */
int txe_gat_aggregate_fwd(const int* rowptr_in, const int* col_src, int n_nodes, const float* ft, long long ld_ft,
                          const float* a_src, const float* a_dst, int ld_a, int H, int D, float attn_slope, float attn_drop_p,
                          unsigned long long seed, int out_mode, float act_slope, float* out, long long ld_out, float* alpha,
                          const float* nx_wa, int nx_kp, const unsigned* nx_mask, float nx_feat_drop_p, float* nx_a12, int npw, void* stream);
/* npw: 0 = the sweep is chosen from the batch (batches of 4,096 nodes and more with four heads and 16-byte rows walk egonets:
 * gat_aggregate_ego_kernel, a workgroup per window of consecutive nodes, every row read once); 1 | 2 = one wave per node with that many
 * nodes per wave; 4 = one wave per node, nodes per wave from the batch size; 3 | 8..32 = the egonet walk, forced (with that many nodes per
 * window; TXE_ERR_ARG if the shape does not fit).  `out` and `alpha` are bit-equal across all of them, nx_a12 within rounding (a
 * parity test compares them).
 * d_pre = gradient w.r.t. the PRE-activation aggregated output.  Writes d_ft [N][H*D], d_a_src/d_a_dst [N][H]
 * (row stride ld_da).  dz_ws: E*H floats of scratch.  n_pad: floats following d_a_dst[v][H-1] in every row that are cleared as
 * well (the zero padding columns of txe_gat_dense_bwd's d_Y operand when d_ft | d_a_src | d_a_dst share one padded row); 0 = none. */
int txe_gat_aggregate_bwd(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst,
                          const int* pos_out, int n_nodes, const float* ft, long long ld_ft, const float* a_src,
                          const float* a_dst, int ld_a, int H, int D, float attn_slope, float attn_drop_p,
                          unsigned long long seed, const float* alpha, const float* d_pre, long long ld_dpre, float* d_ft,
                          long long ld_dft, float* d_a_src, float* d_a_dst, int ld_da, float* dz_ws, int n_pad, void* stream);
int txe_leaky_relu_bwd(const float* d_out, const float* out_act, float slope, long long n, float* d_pre, void* stream);
/* `.mean(1)` over heads of the output layer, model_zoo.py:219 */
int txe_head_mean_fwd(const float* x, int H, int D, long long n_rows, float* y, void* stream);
int txe_head_mean_bwd(const float* dy, int H, int D, long long n_rows, float* dx, void* stream);

/* ---- GCNLayer: model_zoo.py:34-50 and the norm of :157-161 ------------------------------------------------------ */
/* dense part on padded operands (as for GAT): Wp [roundup(Kp,128)][Fop] = weight [Kt][Fo] zero padded, Fop = roundup(Fo,32) =
 * txe_gcn_padded_f; hw / d_hw [N][Fop] (d_hw with zero padding columns).  txe_gcn_dense_bwd writes d_X [N][Kp] columns [c0,Kt)
 * exactly like txe_gat_dense_bwd, dW [Kt][Fo] and dP [vocab][Pd]. */
int txe_gcn_padded_f(int Fo);
int txe_gcn_pack_weights(const float* W, int Kt, int Fo, float* Wp, void* stream);
size_t txe_gcn_dense_ws_bytes(int n_nodes, int Kh, int Pd, int Fo, int vocab);
int txe_gcn_dense_fwd(const float* X, int n_nodes, int Kh, int Pd, const float* Wp, int Fo, float drop_p, const unsigned* mask,
                      float* hw, void* ws, size_t ws_bytes, void* stream);
int txe_gcn_dense_bwd(const float* X, int n_nodes, int Kh, int Pd, const int* pos, int vocab, const float* Wp, int Fo, float drop_p,
                      const unsigned* mask, const float* d_hw, int need_dh, int act_on, float act_slope, float* d_X, float* dW,
                      float* dP, int x_dropped, void* ws, size_t ws_bytes, void* stream);
int txe_gcn_norm(const int* rowptr_in, int n_nodes, float* norm, void* stream);
int txe_gcn_aggregate_fwd(const int* rowptr_in, const int* col_src, int n_nodes, const float* hw, long long ld_hw,
                          const float* norm, const float* bias, int has_act, float act_slope, int F, float* out, long long ld_out,
                          void* stream);
size_t txe_gcn_aggregate_bwd_ws_bytes(int n_nodes, int F);
int txe_gcn_aggregate_bwd(const int* rowptr_out, const int* col_dst, int n_nodes, const float* d_pre, long long ld_dpre,
                          const float* norm, int F, float* d_hw, long long ld_dhw, float* d_bias, void* ws, size_t ws_bytes,
                          void* stream);   /* also zeroes d_hw's columns [F, min(ld_dhw, roundup(F, 32))): txe_gcn_dense_bwd's zero padding */

/* ---- readouts: MeanReadout model_zoo.py:231-232 (pw = NULL), WeightedMeanReadout :240-242 ------------------------- */
int txe_readout_fwd(const int* graph_off, int G, const float* h, long long ld_h, const int* pos, const float* pw, int D,
                    float* hg, float* wsum, void* stream);
int txe_readout_bwd(const int* graph_off, int G, const float* h, long long ld_h, const int* pos, const float* pw, int vocab,
                    int D, const float* hg, const float* wsum, const float* d_hg, float* d_h, long long ld_dh, float* d_pw,
                    float* dpw_ws, void* stream);

/* SumReadout (mode 1), MaxReadout (mode 2), ConcatReadout (mode 3: [G][3D]) -- model_zoo.py:244-276 */
int txe_readout_multi_fwd(const int* graph_off, int G, const float* h, long long ld_h, const int* pos, int D, int mode, float* hg,
                          int* argmax, void* stream);
int txe_readout_multi_bwd(const int* graph_off, int G, const int* pos, int D, int mode, const float* d_hg, const int* argmax,
                          float* d_h, long long ld_dh, void* stream);

/* ---- matchers: BIM model_zoo.py:313, LBM :328 (apply_exp) ------------------------------------------------------- */
int txe_bilinear_project(const float* e1, long long ld_e1, int G, int l, const float* W, int r, float* U, long long ld_u,
                         void* sws, size_t sws_bytes, void* stream);   /* sws: NULL, or txe_gemm_plain_split_ws_bytes(G, r, l) of scratch:
                         the product then runs on the bf16 matrix pipe in fp32 accuracy */
int txe_bilinear_pair_fwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r,
                          const float* W, int apply_exp, float* U, float* s, void* stream);
size_t txe_bilinear_pair_bwd_ws_bytes(int G, int l, int r);
int txe_bilinear_pair_bwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r,
                          const float* W, int apply_exp, const float* U, const float* s, const float* ds, float* d_e1,
                          long long ld_de1, float* d_e2, long long ld_de2, float* dW, void* ws, size_t ws_bytes, void* stream);
/* query-side form of the pairwise match for queries that need no gradient (training: model.py:86, trainer.py:51):
 * forward V = E2 W^T [G][l], s_i = <e1_i, V_i> (exp optionally); backward d_e1_i = dsl_i V_i (elementwise -- no second G-row GEMM),
 * dW = (dsl (.) E1)^T E2.  V is kept for backward. */
int txe_bilinear_query_fwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r, const float* W,
                           int apply_exp, float* V, float* s, void* stream);
/* its two halves, for a caller that launches the projection early (it needs the queries and W only) on another stream */
int txe_bilinear_query_project(const float* e2, long long ld_e2, int G, int l, int r, const float* W, float* V, void* stream);
int txe_bilinear_query_dot(const float* e1, long long ld_e1, const float* V, int G, int l, int apply_exp, float* s, void* stream);
size_t txe_bilinear_query_bwd_ws_bytes(int G, int l, int r);
int txe_bilinear_query_bwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, int G, int l, int r, int apply_exp,
                           const float* V, const float* s, const float* ds, float* d_e1, long long ld_de1, float* dW, void* ws,
                           size_t ws_bytes, void* stream);

/* the pairwise match when the query rows repeat in RUNS: a training batch pairs one query with 1 + negative_size consecutive anchors and
 * data_loaders.py:9-28 stacks that query's row once per pair (trainer.py:46,51 hands the stack to model.py:86).  Qu [U][r] = the
 * distinct rows, run_off [U+1] = first pair of every run (run_off[U] = G): V = Qu W^T [U][l] is U rows instead of G, s_i = <e1_i, V[run(i)]>;
 * backward d_e1_i = dsl_i V[run(i)], dW = S^T Qu with S[u] = sum over run u (pairs in order) of dsl_i e1_i -- K = U instead of G.  Same values
 * as txe_bilinear_query_* on the stacked rows up to the summation order of dW. */
int txe_bilinear_runs_fwd(const float* e1, long long ld_e1, const float* Qu, long long ld_q, const int* run_off, int G, int U, int l, int r,
                          const float* W, int apply_exp, float* V, float* s, void* stream);
size_t txe_bilinear_runs_bwd_ws_bytes(int U, int l, int r);
int txe_bilinear_runs_bwd(const float* e1, long long ld_e1, const float* Qu, long long ld_q, const int* run_off, int G, int U, int l, int r,
                          int apply_exp, const float* V, const float* s, const float* ds, float* d_e1, long long ld_de1, float* dW, void* ws,
                          size_t ws_bytes, void* stream);

/* ... and when the caller only has the reference collate's STACKED matrix E2 [G][r]: txe_rows_find_runs compares every row bit for bit
 * with its predecessor and numbers the runs on the device (run_id [G], run_off [G + 1] with run_off[n_runs] = G, n_runs [1]; no host
 * synchronisation), txe_bilinear_stacked_* are txe_bilinear_runs_* on E2 itself (run u's row = row run_off[u]; V [G][l] and the workspace
 * are sized for G runs, the kernels walk the actual count).  Right for any input; worth it when rows repeat (the caller decides). */
int txe_rows_find_runs(const float* e2, long long ld_e2, int G, int r, int* run_id, int* run_off, int* n_runs, void* stream);
int txe_bilinear_stacked_fwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, const int* run_off, const int* n_runs, int G,
                             int l, int r, const float* W, int apply_exp, float* V, float* s, void* stream);
size_t txe_bilinear_stacked_bwd_ws_bytes(int G, int l, int r);
int txe_bilinear_stacked_bwd(const float* e1, long long ld_e1, const float* e2, long long ld_e2, const int* run_off, const int* n_runs, int G,
                             int l, int r, int apply_exp, const float* V, const float* s, const float* ds, float* d_e1, long long ld_de1,
                             float* dW, void* ws, size_t ws_bytes, void* stream);

/* The same match FOLDED through the encoder's output layer (model.py:86 on model_zoo.py:313/328 behind :227-242 and the last GATLayer):
 * with hg = Z Wf^T (txe_gat_collapse_fwd, hg == NULL) the score is s_i = <Z_i, T[u(i)]>, T[u] = Wf^T (Wm q_u) -- the D x Kp product runs on
 * the U run rows instead of the G graph rows, forward and backward (dZ_i = dsl_i T[u(i)]; dT[u] = sum dsl_i Z_i; dV = dT Wf^T; dWf = V^T dT,
 * handed to txe_gat_collapse_bwd_fused as dw_main; dWm = dV^T Q).  Runs as in txe_bilinear_runs_* (first_row 0, n_runs NULL, Q = the U
 * distinct rows) or txe_bilinear_stacked_* (first_row 1, the count on the device, U = G sizes V [U][l], T / dT [U][Kp], dV [U][l]). */
int txe_bilinear_folded_fwd(const float* Z, long long ld_z, int G, int Kp, const float* Wf, long long ld_wf, int l, const float* Q, long long ld_q,
                            int r, const int* run_off, const int* n_runs, int U, int first_row, const float* Wm, int apply_exp, float* V, float* T,
                            float* s, int stages /* 1: V and T (queries and weights only), 2: the scores (Z), 3: both */,
                            int wf_by_k /* 0: Wf [l][Kp] (GAT packing); 1: Wf [Kp][ld_wf] (GCN packing), dWf then [Kp][l] */,
                            int one_col /* -1, or the column of Z that counts as 1: row one_col of a by-k Wf holds the layer's bias */, void* stream);
int txe_runs_expand(const int* run_off, int U, int G, int* run_id, void* stream);   /* run_id[i] = the run that holds pair i */
int txe_bilinear_folded_bwd(const float* Z, long long ld_z, int G, int Kp, const float* Wf, long long ld_wf, int l, const float* Q, long long ld_q,
                            int r, const int* run_off, const int* n_runs, int U, int first_row, int apply_exp, const float* V, const float* T,
                            const float* s, const float* ds, float* dZ, long long ld_dz, float* dT, float* dV, float* dWm, float* dWf, int wf_by_k,
                            int one_col, void* stream);

/* nn.Linear over the (virtual) concat of two inputs + activation (0 none / 1 relu / 2 tanh): the MLP matcher, model_zoo.py:285-298 */
int txe_linear_fwd(const float* x1, long long ld1, int l, const float* x2, long long ld2, int r, int G, const float* W, const float* b,
                   int O, int act, float* y, void* stream);
size_t txe_linear_bwd_ws_bytes(int G, int l, int r, int O);
int txe_linear_bwd(const float* x1, long long ld1, int l, const float* x2, long long ld2, int r, int G, const float* W, int O, int act,
                   const float* y, const float* dy, float* dx1, long long ld_dx1, float* dx2, long long ld_dx2, float* dW, float* db,
                   void* ws, size_t ws_bytes, void* stream);
/* ---- all-candidate scoring loop: test_fast.py:116-123 / infer.py:95-99.  U = txe_bilinear_project(hg, W) once, then
 * per query block S[q][g] = match(hg[g], Q[q]) for every candidate g. */
int txe_score_block(const float* Q, long long ld_q, int nq, const float* U, long long ld_u, int G, int r, int apply_exp, float* S,
                    long long ld_s, void* ws, size_t ws_bytes, void* sws, size_t sws_bytes, const void* u_packed, void* stream);
/* ws: optional, txe_gemm_tail_ws_bytes() of scratch.  sws (here and in the three entry points below; NULL = the fp32 MFMA): scratch of
 * txe_score_split_ws_bytes(nq, G or n_pos, r) bytes -- the product then runs on the bf16 matrix pipe in fp32 accuracy (txe_gemm_nt_split
 * below: Q and U are packed into sws per call) through the SAME epilogues.  The four entry points compare scores bit for bit among
 * themselves: give all of them a workspace or none.
 * u_packed (NULL = pack here): the candidates' planes, txe_split_pack(U, ld_u, G, r, side 1, ...) into txe_split_packed_bytes(G, r) bytes,
 * made ONCE per candidate set -- the loop over query blocks then packs only its queries, and sws needs txe_split_packed_bytes(nq, r)
 * (rounded up to 256) bytes only.  Same planes, same kernel: bit-identical scores either way. */
size_t txe_score_split_ws_bytes(int nq, int G, int r);

/* fused scoring + ranking: the score tile is compared in the GEMM epilogue and never stored.  counts [pos_off[nq]] int32
 * (zeroed by the caller; shards of candidates add) += #{g : score(q, g) strictly better than thr[j]}; thr[j] = score of query q's j-th
 * true parent as computed by txe_score_block (bit-identical k-order).  txe_rank_finalize: ranks[j] = 1 + counts[j] - (the query's
 * other positives that beat j) -- metric.py:7-31. */
int txe_score_count_block(const float* Q, long long ld_q, int nq, const float* U, long long ld_u, int G, int r, int apply_exp,
                          const int* pos_off, const float* thr, int larger_is_better, int* counts, void* sws, size_t sws_bytes,
                          const void* u_packed, void* stream);
int txe_rank_finalize(const int* pos_off, int nq, const float* thr, const int* counts, int larger_is_better, int* ranks, void* stream);
/* the thresholds themselves: Up [n_pos][r] = the candidate rows of the queries' true parents, query by query (gathered by the caller);
 * thr[j] = match(Q[q], Up[j]) for j in [pos_off[q], pos_off[q+1]) -- the score kernel's own tiles (bit-identical values), but only the
 * tiles along that staircase are computed and only those pairs are stored (test_fast.py:121-123 restricted to rearrange()'s positives) */
int txe_score_positives(const float* Q, long long ld_q, int nq, const float* Up, long long ld_u, int n_pos, int r, int apply_exp,
                        const int* pos_off, float* thr, void* sws, size_t sws_bytes, void* stream);

/* ---- best-k parents of the scoring loop: infer.py:96-106 / test_fast.py:121-131 (`sorted(enumerate(scores), key=...)[:5]`) --------
 * Fused scoring + selection of one query block: no [nq x G] score block is materialised.  Every 128 x 128 tile of the score GEMM
 * (the kernel and k order of txe_score_block: bit-identical scores) leaves each row's k best columns in part_key / part_idx
 * [nq][txe_score_topk_tiles(G)][k] (caller-provided scratch; floor_ws [nq] ints as well: every row's rising selection floor -- a tile
 * skips values below the best k-th key another tile of the row has already secured); txe_topk_merge picks each query's best k: out_idx [nq][k] candidate rows
 * + idx_base (the first row of a candidate shard), best first, -1 where a row has fewer than k candidates; out_key [nq][k] (may be
 * NULL) = the scores, negated when smaller is better.  Order = Python's stable sort: better score first, equal scores by ascending
 * candidate row; NaN ranks last.  1 <= k <= 8.
 * txe_topk_merge alone: best k of `cnt` (key, idx) entries per row (idx == INT_MAX: empty slot) -- also the merge of the per-rank
 * lists of a candidate-sharded loop. */
int txe_score_topk_tiles(int G);
int txe_score_topk_block(const float* Q, long long ld_q, int nq, const float* U, long long ld_u, int G, int r, int apply_exp,
                         int larger_is_better, int k, int idx_base, float* part_key, int* part_idx, int* floor_ws, int* out_idx,
                         float* out_key, void* sws, size_t sws_bytes, const void* u_packed, void* stream);
int txe_topk_merge(const float* keys, const int* idx, int nq, long long cnt, int k, int idx_base, int* out_idx, float* out_key,
                   void* stream);

/* plain dense product on the fp32 MFMA GEMM (tests / micro-benchmarks).  layout 0: C = A[M][K] B[N][K]^T; 1: C = A[M][K] B[K][N];
 * 2: C = A[K][M]^T B[K][N].  splits > 1: `splits` partial products at C + z*M*ldc.  ws/ws_bytes (optional, txe_gemm_tail_ws_bytes):
 * scratch that lets the last, partial round of workgroups be split along k ("tail splitting").
 * route: 0 = the route the model paths take; test bits selecting a bit-equal alternative kernel: 1 = whole rounds on gemm_kernel instead
 * of the persistent kernel, 2 = split-K TN products without the LDS-direct copies, 4 = every eligible split-K product on 128 x 160 tiles.
 * route bit 8 (layout 0, splits 1): the product runs on the bf16 matrix pipe in fp32 accuracy (txe_gemm_nt_split below); ws then holds the
 * packed operands -- txe_gemm_plain_split_ws_bytes(M, N, K) bytes. */
size_t txe_gemm_tail_ws_bytes(void);
size_t txe_gemm_plain_split_ws_bytes(int M, int N, int K);
int txe_gemm_plain(int layout, const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N,
                   int K, int splits, int route, void* ws, size_t ws_bytes, void* stream);

/* ---- rank extraction of the scoring loop: test_fast.py:16-22 + model/metric.py:7-31 (strict inequalities, the
 * query's other positives excluded).  pos_off [nq+1], pos_idx: candidate columns of each query's true parents. */
int txe_rank_block(const float* S, long long ld_s, int nq, int G, const int* pos_off, const int* pos_idx, int* ranks,
                   int larger_is_better, void* ws_unused, void* stream);

/* ---- graph structure: the batched-egonet COO of dgl.batch (data_loaders.py:25; edge order dataset.py:431-435) to the
 * two CSR views the kernels read (stable: in-edges stay in edge-id order). */
size_t txe_build_csr_ws_bytes(int n_nodes, int n_edges);
int txe_build_csr(const int* src, const int* dst, int n_nodes, int n_edges, int* rowptr_in, int* col_src, int* eid_in,
                  int* rowptr_out, int* col_dst, int* pos_out, void* ws, size_t ws_bytes, void* stream);

/* ---- output GATLayer (ONE head) folded behind MeanReadout / WeightedMeanReadout: model_zoo.py:80-104,219,227-242.
 * hg[g] = (sum_{u in g} c_u Xd[u]) W^T with c_u = sum_{v: u->v} w_v alpha'_uv / S_g -- the same arithmetic as projection ->
 * edge-softmax aggregation -> weighted mean, re-associated so that the projection and its dX / dW products run on G graph rows
 * instead of N node rows.  X [N][Kp] / Wp [Fp][Kp] / mask as for txe_gat_dense_*; pw == NULL: MeanReadout.  Forward keeps
 * a12 [N][2], alpha [E], coef [N], wsum [G], gid [N], Z [G][Kp] for backward; d_X has the layout txe_gat_dense_bwd produces. */
size_t txe_gat_collapse_ws_bytes(int n_nodes, int n_edges, int G, int Kh, int Pd, int D, int vocab);
/* ws_bytes >= txe_gat_collapse_ws_bytes + txe_gat_collapse_split_ws_bytes: txe_gat_collapse_fwd forms hg = Z W^T on the bf16 matrix pipe in
 * fp32 accuracy (txe_gemm_nt_split below) */
size_t txe_gat_collapse_split_ws_bytes(int G, int Kh, int Pd, int D);
int txe_gat_collapse_fwd(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst, const int* pos_out,
                         const int* graph_off, int n_nodes, int n_edges, int G, const float* X, int Kh, int Pd, const float* Wp, int D,
                         float feat_drop_p, const unsigned* mask, float attn_slope, float attn_drop_p, unsigned long long seed,
                         const int* pos, const float* pw, float* a12, int a12_ready /* bit 0: a12 holds the logits already; bit 1: hg = Z W^T on
                         the bf16 matrix pipe -- ws_bytes >= txe_gat_collapse_ws_bytes + txe_gat_collapse_split_ws_bytes, else TXE_ERR_WORKSPACE */,
                         float* alpha, float* coef, float* wsum, int* gid, float* Z,
                         float* hg, long long ld_hg /* hg NULL: stop at Z (the consumer folds hg = Z W^T: txe_bilinear_folded_*) */,
                         const float* Tf, const int* zrow, float* e_part /* all NULL, or (with hg NULL): see phases | 512 below */, void* ws,
                         size_t ws_bytes, void* stream);
int txe_gat_collapse_e_tiles(int n_nodes, int G, int Kh, int Pd);   /* floats per node of e_part; 0: this batch cannot form it */
/* the folded matcher's scores from e_part: s_g = [exp] <Z_g, Tf[zrow[g]]> = [exp] (scale / S_g) sum_{u in g} coef_u e_u -- no sweep over Z
 * (masked: the layer's keep mask was applied, i.e. scale = 1 / (1 - feat_drop_p)) */
int txe_gat_collapse_fold_scores(const int* graph_off, int n_nodes, int G, int Kh, int Pd, const float* coef, const float* wsum, const float* e_part,
                                 float feat_drop_p, int masked, int apply_exp, float* s, void* stream);
int txe_gat_collapse_bwd(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst, const int* pos_out,
                         const int* graph_off, int n_nodes, int n_edges, int G, const float* X, int Kh, int Pd, const int* pos, int vocab,
                         const float* Wp, const float* W, const float* attn_l, const float* attn_r, int D, float feat_drop_p,
                         const unsigned* mask, float attn_slope, float attn_drop_p, unsigned long long seed, const float* pw,
                         const float* a12, const float* alpha, const float* coef, const float* wsum, const int* gid, const float* Z,
                         const float* hg, long long ld_hg, const float* d_hg, long long ld_dhg, int act_on, float act_slope, float* d_X, float* dW, float* d_attn_l,
                         float* d_attn_r, float* dP, float* d_pw, void* ws, size_t ws_bytes, void* stream);

/* txe_gat_collapse_bwd FUSED with txe_gat_aggregate_bwd of the GATLayer below it (one HBM sweep instead of three): the
 * gradient of the folded layer's input is formed on the fly from X, dZ and the attention-logit gradients while the layer below's
 * source-side sweep runs, and never stored.  Extra inputs: that layer's projection output Yp [N][ld_yp] = [ft | a1 | a2] (Hp heads x
 * Dp, Hp*Dp == Kh), its attention alpha_p [E][Hp] (destination-CSR order), slope / dropout / seed, the slope of the activation between
 * the layers (1 = none).  Output instead of d_X: d_Yp [N][ld_dyp] = [d_ft | d_a1 | d_a2 | n_pad zeros]; dz_p [E][Hp] scratch.
 * phases: 15 = all of it; 1 | 2 | 4 | 8 = dZ GEMM | dW GEMM partials (independent of 1 and 4: a second stream may run it under the sweeps)
 * | sweeps + first reduction stage | final reductions -- separate calls share the workspace; 8 | 64 with a chain defers the final
 * reductions (see txe_gat_dense_bwd).  | 128 (on EVERY call of one backward pass: the workspace layout depends on it): the dW product
 * runs beside other kernels on a second stream and is cut into at most 2 fat k-slices, which leave those kernels their wave slots.
 * | 256: the caller folded hg = Z W^T into the consumer of Z (txe_bilinear_folded_*; txe_gat_collapse_fwd with hg == NULL stops at Z):
 * `d_hg` IS dZ [G][Kp] (ld_dhg == Kp), hg may be NULL, phases 1 and 2 do not run, and the main part of dW comes from the caller as
 * dw_slices slices [D][Kp] at dw_main (summed in order; 0 slices: none) -- this call adds the attention rows' part.
 * | 512 (with | 256): the <dZ, X> sweep was formed in forward -- txe_gat_collapse_fwd with Tf [runs][Kp], zrow [G] (graph -> its row of Tf) and
 * e_part [N][txe_gat_collapse_e_tiles] given leaves <Tf[zrow[g]], keep X[u]> there; with the folded matcher's dZ[g] = dsl_g Tf[zrow[g]] backward
 * needs only its score gradient m_ds [G], its scores m_s [G], whether it exponentiates (m_exp), and Tf / zrow again: the fused sweep reads its
 * "dZ[g]" rows as Tf[zrow[g]] with dsl_g folded into the node coefficients -- d_hg may be NULL, dZ is never formed.
 * txe_gat_fused_bwd_supported: 1 if the shape qualifies (Hp in {1,2,4}, Hp*Dp % 16 == 0, <= 128 columns behind the feature part). */
int txe_gat_fused_bwd_supported(int Kh, int Pd, int Hp, int Dp);
size_t txe_gat_collapse_bwd_fused_ws_bytes(int n_nodes, int n_edges, int G, int Kh, int Pd, int D, int vocab, int Hp);
int txe_gat_collapse_bwd_fused(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst, const int* pos_out,
                               const int* graph_off, int n_nodes, int n_edges, int G, const float* X, int Kh, int Pd, const int* pos,
                               int vocab, const float* Wp, const float* W, const float* attn_l, const float* attn_r, int D,
                               float feat_drop_p, const unsigned* mask, float attn_slope, float attn_drop_p, unsigned long long seed,
                               const float* pw, const float* a12, const float* alpha, const float* coef, const float* wsum,
                               const int* gid, const float* Z, const float* hg, long long ld_hg, const float* d_hg, long long ld_dhg,
                               float act_slope, const float* Yp, long long ld_yp, int Hp, int Dp, float attn_slope_p,
                               float attn_drop_p_p, unsigned long long seed_p, const float* alpha_p, float* d_Yp, long long ld_dyp,
                               int n_pad, float* dz_p, float* dW, float* d_attn_l, float* d_attn_r, float* dP, float* d_pw, int phases,
                               const float* dw_main, int dw_slices, const float* e_part, const float* m_ds, const float* m_s, int m_exp,
                               const float* Tf, const int* zrow, int* zgid /* [N] scratch */, const int* walk_plan, void* chain, void* ws,
                               size_t ws_bytes, void* stream);
/* walk_plan (optional, NULL = none): the batch's plan from txe_egonet_walk_plan -- what the egonet-walking sweep otherwise works out from
 * the CSR arrays in every workgroup of every step (hub, roles, CSR positions, list order of each graph) done once per batch of graphs:
 * the sweep's staging is then two dependent trips instead of eight.  Same results bit for bit.  plan: txe_egonet_walk_plan_bytes(n_nodes)
 * bytes, 16-byte aligned; depends on the graphs only (both CSR orders, graph offsets), valid as long as they are. */
size_t txe_egonet_walk_plan_bytes(int n_nodes);
int txe_egonet_walk_plan(const int* rowptr_in, const int* col_src, const int* rowptr_out, const int* col_dst, const int* pos_out,
                         const int* graph_off, int n_nodes, int G, int* plan, void* stream);

/* ---- output GCNLayer folded behind MeanReadout / WeightedMeanReadout: model_zoo.py:35-47,139-167,227-242.
 * hg[g] = (sum_{u in g} c_u Xd[u]) W + b with c_u = norm_u sum_{v: u->v} w_v norm_v / S_g (graph constants).  X / Wp / mask as for
 * txe_gcn_dense_*; norm from txe_gcn_norm; forward keeps coef [N], wsum [G], gid [N], Z [G][Kp]. */
size_t txe_gcn_collapse_ws_bytes(int n_nodes, int G, int Kh, int Pd, int Fo, int vocab);
int txe_gcn_collapse_fwd(const int* rowptr_out, const int* col_dst, const int* graph_off, int n_nodes, int G, const float* X, int Kh, int Pd,
                         const float* Wp, int Fo, const float* bias, float drop_p, const unsigned* mask, const float* norm, const int* pos,
                         const float* pw, float* coef, float* wsum, int* gid, float* Z, float* hg, long long ld_hg, void* ws,
                         size_t ws_bytes, void* stream);
int txe_gcn_collapse_bwd(const int* rowptr_in, const int* col_src, const int* graph_off, int n_nodes, int G, const float* X, int Kh, int Pd,
                         const int* pos, int vocab, const float* Wp, int Fo, float drop_p, const unsigned* mask, const float* norm,
                         const float* pw, const float* coef, const float* wsum, const int* gid, const float* Z, const float* d_hg,
                         long long ld_dhg, int act_on, float act_slope, float* d_X, float* dW, float* d_b, float* dP, float* d_pw,
                         int dz_given /* d_hg IS dZ [G][Kp]: no product, dW / d_b not written (txe_bilinear_folded_*, wf_by_k) */, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- egonet construction + batching on device: data_loader/dataset.py:404-437 (_get_subgraph) + dgl.batch (data_loaders.py:25).
 * Taxonomy as parent CSR (par_ptr/par_idx) and child CSR (chd_ptr/chd_idx); anchors [G]; exclude [G] or NULL (query node removed
 * from each egonet's siblings, -1 = none: the positive example of dataset.py:421-424); children beyond `expand` are drawn with
 * replacement from a counter-based hash of (seed, index_base + egonet, draw) -- index_base = position of the batch's first egonet in the
 * caller's whole list, so that a list cut into `-b` chunks (test_fast.py:149-179) draws what the single batch draws.  Step 1 gives node_off [G+1] (node_off[G] = N; E = 2N - G), step 2
 * the node table (ids, pos [N]) and both CSR views in closed form (no sort). */
size_t txe_egonet_ws_bytes(int G);
int txe_egonet_offsets(const int* par_ptr, const int* chd_ptr, const int* chd_idx, const int* anchors, const int* exclude, int G,
                       int expand, unsigned long long seed, int index_base, int* node_off, void* ws, size_t ws_bytes, void* stream);
int txe_egonet_fill(const int* par_ptr, const int* par_idx, const int* chd_ptr, const int* chd_idx, const int* anchors,
                    const int* exclude, int G, int expand, unsigned long long seed, int index_base, const int* node_off, int* ids, int* pos,
                    int* rowptr_in, int* col_src, int* eid_in, int* rowptr_out, int* col_dst, int* pos_out, void* stream);

/* ---- optional per-kernel timing (debug / bench): HIP events on the launch stream around every kernel launch, with the
 * algorithmic work (flops or compulsory bytes) its launcher attributes to it.  Global state (see the conventions at the top); off by
 * default.  txe_profile_get synchronises on record i's events. */
int txe_profile_enable(int on);
int txe_profile_reset(void);
int txe_profile_count(void);
int txe_profile_get(int i, char* name_buf, int buf_len, float* ms, double* work, int* kind);
int txe_profile_stream(int i, void** stream);   /* the hipStream_t record i was launched on */

/* Order two streams of the SAME device: work submitted to `then` after this call starts only when everything submitted to `first`
 * before it has completed (an event without the system-scope fence: no L2 write-back / invalidate in front of `first`'s next kernel).
 * The host-side mirror uses it for every second-stream overlap (taxoexpan_amd/ops.py _order). */
int txe_stream_order(void* first, void* then);
/* device-to-device streaming copy with 16-byte loads / stores (n_bytes % 16 == 0, 16-byte aligned): the copy ceiling bench.py shows the
 * HBM-bound sweeps against, beside the 8 TB/s spec */
int txe_copy_stream(const void* src, void* dst, long long n_bytes, void* stream);

/* model/loss.py:52-57 info_nce_loss = F.cross_entropy(output [B][C], target [B], reduction="sum") on the [queries][1 + negatives]
 * regrouping of trainer.py:52-56, together with its gradient:  loss[0] = sum_b (logsumexp(x_b) - x_b[target_b]),
 * d_x[b][c] = softmax(x_b)[c] - [c == target_b].  target NULL = all zeros (what trainer.py:53 passes). */
int txe_info_nce(const float* x, long long ld_x, int B, int Cc, const long long* target, float* loss, float* d_x, long long ld_dx,
                 void* stream);

/* trainer.py:61 `self.optimizer.step()` for torch.optim.Adam (config.mag.json:66-73: lr 1e-3, weight_decay 0, amsgrad true): the whole
 * parameter set in one launch.  params / grads / exp_avg / exp_avg_sq / max_exp_avg_sq are HOST arrays of n_tensors DEVICE pointers
 * (dense fp32, numel[t] elements); max_exp_avg_sq == NULL selects plain Adam; `step` >= 1 is the count of this update.
 *   g' = g + weight_decay p;  m = lerp(m, g', 1-beta1);  v = beta2 v + (1-beta2) g'^2;  vmax = max(vmax, v)
 *   p -= lr / (1-beta1^step) * m / (sqrt(vmax) / sqrt(1-beta2^step) + eps) */
int txe_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  float* const* max_exp_avg_sq, const long long* numel, double lr, double beta1, double beta2, double eps,
                  double weight_decay, long long step, void* stream);

/* ---- fp32 products on the bf16 matrix pipe (replaces the fp32-MFMA route of model_zoo.py:83 `self.fc(...)` for the first
 * layer's projection).  A packed operand holds every fp32 element as the EXACT sum of three bf16 numbers (three planes, stored as 1-KB
 * MFMA fragments: csrc/txe_gemm_split.h); txe_gemm_nt_split forms C [M][N] = A [M][K] B[N][K]^T from six of the nine plane products with
 * fp32 accumulation -- the dropped terms are a quarter of an fp32 multiply's own rounding error in the root mean square (at most twice it).
 * side 0 = the operand whose rows are C's rows, side 1 = the operand whose rows are C's columns.
 * The whole fp32 domain is covered (csrc/txe_gemm_split.h): a packed fragment that holds +-Inf, NaN or |x| >= 2^120 is stored raw with a
 * NaN marker plane, the product kernels check their accumulators after the k-loop and a tile that is not finite recomputes itself with fp32
 * FMAs over the exactly decoded operands -- the same entries are NaN / +Inf / -Inf as in an IEEE fp32 product (tests/test_gpu_split_gemm.py
 * *_on_the_whole_fp32_domain), at scalar speed for those tiles only.  Nonzero |x| < 2^-100 is carried to 2^-126 ABSOLUTE (flush-to-zero
 * semantics, as GPU fp32 units treat subnormals: such elements are routine in gradients and stay on the fast path). */
size_t txe_split_packed_bytes(int rows, int cols);
int txe_split_pack(const float* src, long long ld, int rows, int cols, int side, void* packed, void* stream);   /* side 2 / 3: side 0 / 1 of
    a matrix given as its transpose, src [cols][ld >= rows] */
int txe_gemm_nt_split(const void* A_packed, const void* B_packed, int M, int N, int K, float* C, long long ldc, void* stream);

/* The TN form (weight gradients, model_zoo.py:83 backward: dW = d_Y^T X over the nodes): part[z][M][ldc] = A[rows of slice z]^T B[same rows],
 * z < S, slices of ksplit rows (a multiple of 16).  A [n_rows][lda] is fp32 (split in the product's loader), B comes packed
 * contraction-major by txe_split_pack_t (cols % 4 == 0, 16-byte aligned rows; the last 160-column tile is zero-filled).  M % 128 == 0; lda % 4 == 0, A 16-byte aligned and
 * n_rows * lda * 4 < 2^31 (32-bit byte offsets), else TXE_ERR_ARG -- txe_gat_dense_bwd falls back to the fp32 MFMA by itself. */
size_t txe_split_packed_t_bytes(int rows, int cols);
int txe_split_pack_t(const float* src, long long ld, int rows, int cols, void* packed, void* stream);
int txe_gemm_tn_split(const float* A, long long lda, int M, const void* B_packed_t, int N, int n_rows, int S, int ksplit, float* part,
                      long long ldc, long long split_stride, void* stream);

/* host-side evaluation of the counter-based dropout hash the kernels inline (uniform in [0,1)); keep = u >= p */
float txe_dropout_uniform_host(unsigned long long seed, unsigned long long idx);
unsigned txe_dropout_mask_word_host(unsigned long long seed, unsigned long long word_index, float p);

#ifdef __cplusplus
}
#endif
#endif /* TXE_H */
