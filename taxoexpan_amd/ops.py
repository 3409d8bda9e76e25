"""torch.autograd bindings of the libtxe kernels (include/txe.h).

torch is plumbing here: it owns device memory (caching allocator), the current HIP stream and autograd's tape;
every number is produced by the hand-written HIP kernels.  There is no CPU path -- host tensors raise.
"""
import weakref

import torch

from . import _lib
from ._lib import call, ptr, pure

LEAKY_SLOPE = 0.01  # F.leaky_relu default, the only activation model.py:25-41 passes


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("taxoexpan_amd: tensors must live on the MI355X (no CPU fallback exists); "
                               "got a host tensor")


def _f32(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _rows(t):
    """2-D fp32 tensor with unit column stride; returns (tensor, ld)."""
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t, t.stride(0)


_BACKWARD_TWICE = ("taxoexpan_amd: backward through this propagation stack a second time -- its saved activations (several hundred MB per "
                   "batch) are released by the first backward; run the forward again (retain_graph=True is not supported here)")
# Alternative routes to the same numbers, kept because a parity test compares each with the default one.  Plain module attributes: tests
# monkeypatch them, tests/conftest.py's TXE_TEST_ROUTE sets one for a whole run.  (The library itself reads no environment variable.)
_NO_FUSED_LOGITS = False    # the folded layer's attention logits by their own sweep instead of the aggregation's epilogue
_NO_TABLE_SWEEP = False     # table rows materialised (txe_gather_add_rows) instead of formed inside the sweep
_NO_SIDE_STREAM = False     # everything on the caller's stream
_NO_MATCH_FOLD = False      # the graph vector hg = Z W^T is always formed (never folded into the bilinear matcher's run products)
_NO_FOLD_EDOT = False       # the folded matcher's T does not ride in the Z sweep: backward runs its <dZ, X> sweep
_NO_FUSED_BWD = False       # the folded layer's backward as the unfused chain (d_X' materialised)
_NO_QUERY_RUNS = False      # stacked query rows always take the GEMM form of the bilinear match
_NO_SPLIT_GEMM = False      # the first layer's projection on the fp32 MFMA instead of the bf16 pipe's six plane products
_NO_TAIL_CHAIN = False      # every layer's last reduction launch in place instead of chained into the bottom layer's
_FWD_SWEEP = 0              # txe_gat_aggregate_fwd's npw argument (0 = chosen from the batch; tools/kt_quick.py sets others)
_NO_VIRTUAL_X = False       # a first layer's input X = dropout([h | Emb[pos]]) is written by the preparation launch and read back by the packs
_NO_WALK_PLAN = False       # the egonet-walking backward sweep works the graphs' shapes out of the CSR arrays in every workgroup (no per-batch plan)
_NO_EGO_WALK = False        # the forward sweep runs one wave per node and the fused backward sweep fetches X'[v] per out-edge, instead of walking egonets
_I32_MEMO = {}       # id(source tensor) -> (weakref, version, device, int32 copy): `pos` is converted once per batch, not once per module


def _i32(t, device):
    if t is None:
        return None
    if t.dtype == torch.int32 and t.device == device:
        return t.contiguous()
    key = id(t)
    hit = _I32_MEMO.get(key)
    if hit is not None and hit[0]() is t and hit[1] == t._version and hit[2] == device:
        return hit[3]
    out = t.to(device=device, dtype=torch.int32).contiguous()
    if len(_I32_MEMO) > 64:
        for k in [k for k, v in _I32_MEMO.items() if v[0]() is None]:
            del _I32_MEMO[k]
        if len(_I32_MEMO) > 64:
            _I32_MEMO.clear()
    _I32_MEMO[key] = (weakref.ref(t), t._version, device, out)
    return out


def _empty(shape, ref, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=ref.device)


def _ws(nbytes, ref):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=ref.device)


def new_seed():
    """64-bit dropout seed drawn from torch's CPU generator (so torch.manual_seed makes runs repeatable)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


ROUTES = {}           # kind -> the route the LAST call of that kind took ('match': _Bilinear.forward; 'stack' / 'stack_bwd': the propagation
                      # stack's cfg.final (+ '+edot') and its backward; 'fold': the folded matcher's forward) -- tests assert on it, bench.py
                      # reports it; debug_capture() additionally collects every note of its block in `routes`


def note_route(kind, name):
    ROUTES[kind] = name
    if _CAPTURE is not None:
        _CAPTURE.routes.append((kind, name))


_GRAD_READY = None    # scoring.overlapped_gradient_allreduce: called as (layer index, [parameter gradients]) the moment a layer's are done
_GRAD_FLUSH = None    # ... and once before the stack returns its gradients to autograd
_CAPTURE = None       # debug_capture(): list that receives (csr, cfg, per-layer states) of every stack forward


class _CaptureList(list):
    """debug_capture()'s list of stack forwards, plus `.routes`: every (kind, route) noted inside the block, in order"""

    def __init__(self):
        super().__init__()
        self.routes = []


class debug_capture:
    """`with ops.debug_capture() as runs:` -- every GAT / GCN stack forward inside the block appends (csr, cfg, states): the per-layer
    buffers of the fused stack (X = padded layer input, Y = projection output, alpha [E, H] in destination-CSR order, cl = the folded
    output layer's (a12, alpha, coef, wsum, gid, Z, hg)).  Parity tests read the intermediates the reference exposes per layer
    (model_zoo.py:90-95) from here; nothing is copied and nothing changes in the computation."""

    def __enter__(self):
        global _CAPTURE
        self._prev, _CAPTURE = _CAPTURE, _CaptureList()
        return _CAPTURE

    def __exit__(self, *exc):
        global _CAPTURE
        _CAPTURE = self._prev
        return False


def apply_stack(fn, csr, cfg, *args):
    """fn.apply with the caller's grad mode recorded in cfg: inside Function.forward grad mode is always off and needs_input_grad
    only mirrors requires_grad, so this is how a no_grad pass (evaluation) avoids keeping the backward state -- and may take the
    table-projection path of GatheredRows."""
    cfg.grad_enabled = torch.is_grad_enabled()
    return fn.apply(csr, cfg, *args)


# ================================================================================================================
# Node features that are rows of a taxonomy feature table (SURVEY 8f-2 "dedup by _id")
# ================================================================================================================
class GatheredRows:
    """x[v] = table[index[v]], kept symbolic.  The batched egonets of an evaluation repeat every taxonomy node many times (MAG-Full:
    1.1 M batch nodes over 431 k taxonomy nodes); without dropout the first layer's projection depends only on (taxonomy node,
    position), so PGAT / PGCN in eval mode project the TABLE once (`projection_cache()` keeps it across the chunks of one evaluation)
    and gather.  Every other consumer sees the ordinary [N, d] tensor (materialised on first use)."""

    def __init__(self, table, index):
        self.table, self.index = table, index
        self._tensor = None

    shape = property(lambda self: torch.Size((self.index.shape[0], self.table.shape[1])))
    device = property(lambda self: self.table.device)
    dtype = property(lambda self: self.table.dtype)
    is_cuda = property(lambda self: self.table.is_cuda)
    requires_grad = False

    def dim(self):
        return 2

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def to(self, *args, **kwargs):
        t = self.table.to(*args, **kwargs)
        return self if t is self.table else GatheredRows(t, self.index.to(t.device))

    def tensor(self):
        if self._tensor is None:
            self._tensor = self.table.index_select(0, self.index.long())
        return self._tensor

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]

    def __len__(self):
        return int(self.index.shape[0])

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        owner = getattr(func, "__self__", None)
        if isinstance(owner, type) and issubclass(owner, torch.autograd.Function) and getattr(owner, "accepts_gathered_rows", False):
            with torch._C.DisableTorchFunctionSubclass():       # our own encoders take the symbolic form as it is
                return func(*args, **(kwargs or {}))
        un = lambda a: a.tensor() if isinstance(a, GatheredRows) else a
        return func(*tree_map(un, tuple(args)), **tree_map(un, dict(kwargs or {})))

    def __add__(self, other):
        return self.tensor() + other

    __radd__ = __add__

    def __mul__(self, other):
        return self.tensor() * other

    __rmul__ = __mul__

    def __repr__(self):
        return f"GatheredRows(table={tuple(self.table.shape)}, rows={int(self.index.shape[0])})"


_PROJ_CACHE = None


class projection_cache:
    """`with projection_cache(expected_rows):` -- table projections (first-layer W applied to a whole feature table) are reused by
    every forward inside the block.  Only for a scope in which the weights do not change (one evaluation pass).  expected_rows =
    how many batch nodes the pass will encode in total: the table is projected only if it has fewer rows than that (or than the
    batch at hand)."""

    def __init__(self, expected_rows=0):
        self.expected_rows = int(expected_rows)

    def __enter__(self):
        global _PROJ_CACHE
        self._prev, _PROJ_CACHE = _PROJ_CACHE, ({} if _PROJ_CACHE is None else _PROJ_CACHE)
        _PROJ_CACHE["expected_rows"] = max(_PROJ_CACHE.get("expected_rows", 0), self.expected_rows)
        return self

    def __exit__(self, *exc):
        global _PROJ_CACHE
        _PROJ_CACHE = self._prev
        return False


def _use_table(h, need, feat_p):
    """project the table instead of the batch?  only without gradients / dropout, and when it is less work (or already cached)"""
    return (isinstance(h, GatheredRows) and not need and feat_p == 0.0 and h.table.is_cuda and h.table.dim() == 2
            and h.table.dtype == torch.float32 and h.table.is_contiguous()
            and h.table.shape[0] <= max(h.index.shape[0], 0 if _PROJ_CACHE is None else _PROJ_CACHE.get("expected_rows", 0)))


def _gat_table_projection(st, src):
    """(T [n_table, Fp], T2 [vocab, Fp] or None): the packed first-layer weights applied to every row of the feature table and to the
    position-embedding rows -- features, a1 and a2 columns alike (they are all linear in the input)."""
    key = ("gat", src.table.data_ptr(), tuple(src.table.shape), st.W.data_ptr(), st.al.data_ptr(), None if st.P is None else st.P.data_ptr())
    if _PROJ_CACHE is not None and key in _PROJ_CACHE:
        return _PROJ_CACHE[key]
    tab = src.table
    n_tab, Kh, Kp, Fp = tab.shape[0], st.Kh, st.Kp, st.Fp
    s = _lib.stream_ptr()
    Kt = pure("txe_gat_padded_k", Kh, 0)
    Xt = _empty((n_tab, Kt), tab)
    call("txe_gat_build_x", ptr(tab), tab.stride(0), n_tab, Kh, None, None, 0, ptr(Xt), s)
    Wp = _empty((Fp, Kp), tab)
    call("txe_gat_pack_weights", ptr(st.W), ptr(st.al), ptr(st.ar), st.H, st.D, Kh + st.Pd, ptr(Wp), s)
    tws = _tail_ws(tab)
    T = _empty((n_tab, Fp), tab)
    # the padding columns [Kh, Kt) of Xt are zero, so whatever Wp holds there (position columns) does not contribute
    if _NO_SPLIT_GEMM:
        call("txe_gemm_plain", 0, ptr(Xt), Kt, ptr(Wp), Kp, ptr(T), Fp, n_tab, Fp, min(Kt, Kp), 1, 0, ptr(tws), tws.numel(), s)
    else:                                               # the table's projection on the bf16 pipe (route bit 8)
        wsb = pure("txe_gemm_plain_split_ws_bytes", n_tab, Fp, min(Kt, Kp))
        sws = _ws(wsb, tab)
        call("txe_gemm_plain", 0, ptr(Xt), Kt, ptr(Wp), Kp, ptr(T), Fp, n_tab, Fp, min(Kt, Kp), 1, 8, ptr(sws), wsb, s)
    T2 = None
    if st.Pd > 0:
        T2 = _empty((st.P.shape[0], Fp), tab)
        call("txe_gemm_plain", 0, ptr(st.P), st.Pd, ptr(Wp) + 4 * Kh, Kp, ptr(T2), Fp, st.P.shape[0], Fp, st.Pd, 1, 0, None, 0, s)
    if _PROJ_CACHE is not None:
        _PROJ_CACHE[key] = (T, T2, Wp, src.table, st.W)          # (operands kept alive: the key holds their addresses)
        return _PROJ_CACHE[key]
    return T, T2


def _gcn_table_projection(st, src):
    """(T [n_table, Fop], T2 [vocab, Fop] or None): the first GCNLayer's weight applied to every table row / position-embedding row"""
    key = ("gcn", src.table.data_ptr(), tuple(src.table.shape), st.W.data_ptr(), None if st.P is None else st.P.data_ptr())
    if _PROJ_CACHE is not None and key in _PROJ_CACHE:
        return _PROJ_CACHE[key]
    tab = src.table
    n_tab, Kh, Fop = tab.shape[0], st.Kh, st.Fop
    s = _lib.stream_ptr()
    Kt = pure("txe_gat_padded_k", Kh, 0)
    Xt = _empty((n_tab, Kt), tab)
    call("txe_gat_build_x", ptr(tab), tab.stride(0), n_tab, Kh, None, None, 0, ptr(Xt), s)
    kp128 = (st.Kp + 127) // 128 * 128
    Wp = _empty((kp128, Fop), tab)                 # [Kh + Pd (padded)][Fop]: feature rows first, then the position rows
    call("txe_gcn_pack_weights", ptr(st.W), Kh + st.Pd, st.Fo, ptr(Wp), s)
    tws = _tail_ws(tab)
    T = _empty((n_tab, Fop), tab)
    call("txe_gemm_plain", 1, ptr(Xt), Kt, ptr(Wp), Fop, ptr(T), Fop, n_tab, Fop, min(Kt, kp128), 1, 0, ptr(tws), tws.numel(), s)
    T2 = None
    if st.Pd > 0:
        T2 = _empty((st.P.shape[0], Fop), tab)
        call("txe_gemm_plain", 1, ptr(st.P), st.Pd, ptr(Wp) + 4 * Kh * Fop, Fop, ptr(T2), Fop, st.P.shape[0], Fop, st.Pd, 1, 0, None, 0, s)
    if _PROJ_CACHE is not None:
        _PROJ_CACHE[key] = (T, T2, Wp, src.table, st.W)
        return _PROJ_CACHE[key]
    return T, T2


# ================================================================================================================
# GAT stack (PGAT / GAT / a single GATLayer)
# ================================================================================================================
class GATConfig:
    """static description of a stack of GATLayers (model_zoo.py:52-114,169-220)"""

    def __init__(self, heads, out_dims, pos_dims, vocab, attn_slope, act_slope, feat_p, attn_p, final, seed):
        self.heads, self.out_dims, self.pos_dims, self.vocab = list(heads), list(out_dims), list(pos_dims), vocab
        self.attn_slope, self.act_slope = float(attn_slope), act_slope
        self.feat_p, self.attn_p = float(feat_p), float(attn_p)
        self.final = final          # 'mean' (PGAT/GAT: .mean(1) over heads of the last layer) | 'none' (GATLayer: N x H x D)
        self.seed = int(seed)
        self.n_layers = len(self.heads)


def dropout_mask(n_rows, n_cols, p, seed, ref):
    """keep-bit mask (int32 words [n_rows, ceil(n_cols/32)]) of nn.Dropout(p) over an [n_rows, n_cols] operand, or None"""
    if p <= 0.0:
        return None
    mask = torch.empty((n_rows, (n_cols + 31) // 32), dtype=torch.int32, device=ref.device)
    call("txe_dropout_mask", n_rows, n_cols, p, seed, ptr(mask), _lib.stream_ptr())
    return mask


_tail_ws_cache = {}


def _tail_ws(ref):
    """persistent GEMM tail-splitting scratch per (device, stream): reuse is stream-ordered and the contents never outlive one
    GEMM + its fix-up kernel, so two streams (or threads on their own streams) must not share a buffer"""
    key = (ref.device.index, torch.cuda.current_stream(ref.device).cuda_stream)
    t = _tail_ws_cache.get(key)
    if t is None:
        t = torch.empty(pure("txe_gemm_tail_ws_bytes"), dtype=torch.uint8, device=ref.device)
        _tail_ws_cache[key] = t
    return t


class _GatLayerState:
    __slots__ = ("X", "Wp", "mask", "Y", "alpha", "W", "al", "ar", "P", "Kh", "Pd", "Kp", "Fp", "H", "D", "seed", "cl", "prepared", "x_dropped", "Xt",
                 "vx")      # vx: X is NOT stored (a first layer on the bf16 pipe: the packs form dropout([h | Emb[pos]]) themselves)


def _virtual_x_ok(st, h, ld_h, N, need, first_is_folded):
    """may a FIRST layer's input stay unwritten?  Its only readers must be the two packs of the bf16-pipe products: the projection
    (txe_gat_dense_fwd_split_src) and, with a backward pass to come, the weight gradient's contraction-major form (Xt)."""
    if _NO_SPLIT_GEMM or _NO_VIRTUAL_X or first_is_folded or N == 0 or not torch.is_tensor(h) or h.dtype != torch.float32:
        return False
    if pure("txe_gat_dense_split_ws_bytes", N, st.Kh, st.Pd, st.H, st.D) == 0:
        return False
    return (not need) or pure("txe_gat_dense_split_xt_bytes", N, st.Kh, st.Pd, st.H, st.D) > 0


def _x_dropped_ok(cfg, states, l, collapse):
    """may layer l's input be stored dropped?  Not the folded output layer (its sweeps apply the mask themselves); a layer above the
    first needs the aggregation below to drop what it writes: 16-byte rows, H <= 4 (the fast kernel family), an activation between."""
    L = len(states)
    if collapse and l == L - 1:
        return False
    if l == 0:
        return True
    sp = states[l - 1]
    return sp.D % 4 == 0 and states[l].Kp % 4 == 0 and sp.H <= 4


def _gat_layers_prepare(items, feat_p):
    """_gat_layer_prepare for several layers of a stack in ONE launch (txe_gat_layers_prepare): items = [(st, h, ld_h, pos, dropped)],
    st.X allocated.  A layer's preparation never depends on the layer below's output, so the whole stack is prepared before its first
    GEMM.  dropped: X is written with the feature dropout already applied (a first layer on raw features whose X only GEMMs read)."""
    import ctypes
    descs = (_lib.GatPrepareDesc * len(items))()
    for d, (st, h, ld_h, pos, dropped) in zip(descs, items):
        N = st.X.shape[0]
        st.vx = getattr(st, "vx", False)
        st.Wp = _empty((st.Fp, st.Kp), st.X)
        st.mask = torch.empty((N, (st.Kh + st.Pd + 31) // 32), dtype=torch.int32, device=st.X.device) if feat_p > 0.0 else None
        d.h, d.ld_h, d.n_nodes, d.Kh, d.pos, d.P, d.Pd, d.X = ptr(h), ld_h, N, st.Kh, ptr(pos), ptr(st.P), st.Pd, (None if st.vx else ptr(st.X))
        d.W, d.attn_l, d.attn_r, d.H, d.D, d.Wp = ptr(st.W), ptr(st.al), ptr(st.ar), st.H, st.D, ptr(st.Wp)
        d.feat_drop_p, d.seed, d.mask = feat_p, st.seed, ptr(st.mask)
        st.x_dropped = bool(dropped and feat_p > 0.0)
        d.x_dropped = int(st.x_dropped)
        st.prepared = True
    call("txe_gat_layers_prepare", ctypes.cast(descs, ctypes.c_void_p), len(items), _lib.stream_ptr())


def _gat_layer_prepare(st, h, ld_h, pos, feat_p):
    """layer input X = [h | Emb[pos] | 0] (h == None: the producer already wrote the feature columns), packed weights, keep mask"""
    if getattr(st, "prepared", False):
        return
    N = st.X.shape[0]
    s = _lib.stream_ptr()
    st.Wp = _empty((st.Fp, st.Kp), st.X)
    st.mask = torch.empty((N, (st.Kh + st.Pd + 31) // 32), dtype=torch.int32, device=st.X.device) if feat_p > 0.0 else None
    call("txe_gat_layer_prepare", ptr(h), ld_h, N, st.Kh, ptr(pos), ptr(st.P), st.Pd, ptr(st.X), ptr(st.W), ptr(st.al), ptr(st.ar),
         st.H, st.D, ptr(st.Wp), feat_p, st.seed, ptr(st.mask), s)


def _gat_collapse_fwd(csr, st, h, ld_h, pos, rpos, pw, feat_p, attn_p, attn_slope, a12=None, z_only=False, fold_job=None, link=None):
    """output layer (one head) folded behind the weighted-mean readout: hg [G, D] (txe_gat_collapse_fwd).
    a12 given: the layer is already prepared and the previous layer's aggregation has formed its attention logits.
    z_only: stop at Z [G, Kp] (hg = Z W^T is left to the consumer: FoldedGraphLinearFunction / BilinearFoldedRunsFunction)."""
    N, G, E = st.X.shape[0], csr.n_graphs, csr.n_edges
    ready = a12 is not None
    if not ready:
        _gat_layer_prepare(st, h, ld_h, pos, feat_p)
        a12 = _empty((max(N, 1), 2), st.X)
    alpha, coef = _empty((max(E, 1),), st.X), _empty((max(N, 1),), st.X)
    wsum, Z, hg = _empty((max(G, 1),), st.X), _empty((max(G, 1), st.Kp), st.X), (None if z_only else _empty((G, st.D), st.X))
    gid = torch.empty(max(N, 1), dtype=torch.int32, device=st.X.device)
    wsb = pure("txe_gat_collapse_ws_bytes", N, E, G, st.Kh, st.Pd, st.D, 8)
    split_hg = not z_only and G > 0 and not _NO_SPLIT_GEMM
    if split_hg:                                         # room for Z and the weight rows as packed planes: hg = Z W^T on the bf16 pipe
        wsb += pure("txe_gat_collapse_split_ws_bytes", G, st.Kh, st.Pd, st.D)
    ws = _ws(wsb, st.X)
    Tf = zrow = e_part = None
    if z_only and fold_job is not None and link is not None and N > 0 and G > 0 and not _NO_FOLD_EDOT:
        nt = pure("txe_gat_collapse_e_tiles", N, G, st.Kh, st.Pd)
        fw = fold_job(st.Wp, st.D) if nt > 0 else None       # the matcher's runs, V and T, formed now: T rides in the Z sweep
        if fw is not None:
            Tf, zrow, e_part = fw["T"], _fold_job_run_ids(fw, G, st.X), _empty((N, nt), st.X)
            link.fwd, link.e_part = fw, e_part
            # (what the matcher's forward needs to sum its scores from e_part: txe_gat_collapse_fold_scores)
            fw["score"] = (csr.graph_off, N, G, st.Kh, st.Pd, coef, wsum, feat_p, int(st.mask is not None and feat_p > 0.0))
    call("txe_gat_collapse_fwd", ptr(csr.rowptr_in), ptr(csr.col_src), ptr(csr.rowptr_out), ptr(csr.col_dst), ptr(csr.pos_out),
         ptr(csr.graph_off), N, E, G, ptr(st.X), st.Kh, st.Pd, ptr(st.Wp), st.D, feat_p, ptr(st.mask), attn_slope, attn_p, st.seed + 1,
         ptr(rpos), ptr(pw), ptr(a12), int(ready) | (2 if split_hg else 0), ptr(alpha), ptr(coef), ptr(wsum), ptr(gid), ptr(Z), ptr(hg), st.D, ptr(Tf), ptr(zrow),
         ptr(e_part), ptr(ws), wsb, _lib.stream_ptr())
    st.cl = (a12, alpha, coef, wsum, gid, Z, hg)
    return Z if z_only else hg


def _gat_collapse_bwd(csr, st, pos, rpos, pw, vocab, feat_p, attn_p, attn_slope, d_hg, act_on, act_slope):
    N, G, E = st.X.shape[0], csr.n_graphs, csr.n_edges
    a12, alpha, coef, wsum, gid, Z, hg = st.cl
    d_hg, ld = _rows(d_hg)
    dW, dal, dar = torch.empty_like(st.W), torch.empty_like(st.al), torch.empty_like(st.ar)
    dP = torch.empty_like(st.P) if st.P is not None else None
    d_pw = torch.empty_like(pw) if pw is not None else None
    d_X = _empty((N, st.Kp), st.X)
    v = max(vocab, pw.numel() if pw is not None else 0)
    wsb = pure("txe_gat_collapse_ws_bytes", N, E, G, st.Kh, st.Pd, st.D, max(v, 8))
    ws = _ws(wsb, st.X)
    call("txe_gat_collapse_bwd", ptr(csr.rowptr_in), ptr(csr.col_src), ptr(csr.rowptr_out), ptr(csr.col_dst), ptr(csr.pos_out),
         ptr(csr.graph_off), N, E, G, ptr(st.X), st.Kh, st.Pd, ptr(pos if pos is not None else rpos), v, ptr(st.Wp), ptr(st.W),
         ptr(st.al), ptr(st.ar), st.D, feat_p, ptr(st.mask), attn_slope, attn_p, st.seed + 1, ptr(pw), ptr(a12), ptr(alpha), ptr(coef),
         ptr(wsum), ptr(gid), ptr(Z), ptr(hg), st.D, ptr(d_hg), ld, int(act_on), act_slope if act_slope else 1.0, ptr(d_X), ptr(dW), ptr(dal), ptr(dar), ptr(dP),
         ptr(d_pw), ptr(ws), wsb, _lib.stream_ptr())
    return d_X, dW, dal, dar, dP, d_pw


def _gat_layer_fwd(csr, st, h, ld_h, pos, out, ld_out, feat_p, attn_p, attn_slope, out_mode, act_slope, save, nxt=None, out_drop=None):
    """st.X is pre-allocated [N, Kp]; h != None copies the raw features in, h == None means the producer already wrote them.
    nxt = (prepared state of the next, folded one-head layer, a12 buffer): its attention logits ride in the aggregation's epilogue."""
    H, D, Kh, Pd, Kp, Fp = st.H, st.D, st.Kh, st.Pd, st.Kp, st.Fp
    F = H * D
    s = _lib.stream_ptr()
    if isinstance(h, GatheredRows):            # eval-mode first layer on table rows: project the table, gather
        N = h.index.shape[0]
        T, T2 = _gat_table_projection(st, h)[:2]
        nx_kp = nxt[0].Kp if nxt is not None else 0
        if (T2 is not None and pos is not None and not save and not _NO_TABLE_SWEEP and attn_p == 0.0
                and pure("txe_gat_aggregate_table_supported", H, D, Fp, T2.shape[0], nx_kp) == 1):
            # the projected rows T[id] + T2[pos] are formed inside the sweep: no [N, Fp] round trip through HBM
            st.Y = st.alpha = None
            call("txe_gat_aggregate_table_fwd", ptr(csr.rowptr_in), ptr(csr.col_src), N, ptr(T), Fp, ptr(_i32(h.index, T.device)), ptr(T2),
                 ptr(pos), T2.shape[0], H, D, attn_slope, out_mode, act_slope, ptr(out), ld_out,
                 *((ptr(nxt[0].Wp) + 4 * nxt[0].D * nxt[0].Kp, nx_kp, ptr(nxt[1])) if nxt is not None else (None, 0, None)),
                 4 if _NO_EGO_WALK else (_FWD_SWEEP if H == 4 and D % 4 == 0 else 0), s)
            return
        st.Y = _empty((N, Fp), T)
        call("txe_gather_add_rows", ptr(T), Fp, ptr(_i32(h.index, T.device)), ptr(T2), Fp, ptr(pos) if T2 is not None else None, N, Fp,
             ptr(st.Y), Fp, s)
    else:
        N = st.X.shape[0]
        _gat_layer_prepare(st, h, ld_h, pos, feat_p)
        st.Y = _empty((N, Fp), st.X)
        tws = _tail_ws(st.X)
        dropped = getattr(st, "x_dropped", False)
        if (dropped or feat_p == 0.0) and not _NO_SPLIT_GEMM:      # X is a plain operand: fp32-accurate product on the bf16 pipe
            wsb = pure("txe_gat_dense_split_ws_bytes", N, Kh, Pd, H, D)
            sws = _ws(wsb, st.X)
            xtb = pure("txe_gat_dense_split_xt_bytes", N, Kh, Pd, H, D) if save else 0
            st.Xt = _ws(xtb, st.X) if xtb else None        # X packed contraction-major: the backward pass's weight gradient reads it
            if getattr(st, "vx", False):                   # X was never written: the packs read h, the position table and the mask
                call("txe_gat_dense_fwd_split_src", ptr(h), ld_h, ptr(pos), ptr(st.P), ptr(st.mask), feat_p if st.mask is not None else 0.0,
                     N, Kh, Pd, ptr(st.Wp), H, D, ptr(st.Xt), ptr(st.Y), ptr(sws), wsb, s)
            else:
                call("txe_gat_dense_fwd_split", ptr(st.X), N, Kh, Pd, ptr(st.Wp), H, D, None, None, ptr(st.Xt), ptr(st.Y), ptr(sws), wsb, s)
            note_route("proj", "bf16x6")
        else:
            call("txe_gat_dense_fwd", ptr(st.X), N, Kh, Pd, ptr(st.Wp), H, D, 0.0 if dropped else feat_p, None if dropped else ptr(st.mask), ptr(st.Y),
                 ptr(tws), tws.numel(), s)
            note_route("proj", "fp32")
        _launch_pending_prefetch()
    st.alpha = _empty((max(csr.n_edges, 1), H), st.Y) if save else None
    call("txe_gat_aggregate_fwd", ptr(csr.rowptr_in), ptr(csr.col_src), N, ptr(st.Y), Fp, ptr(st.Y) + 4 * F, ptr(st.Y) + 4 * (F + H), Fp,
         H, D, attn_slope, attn_p, st.seed + 1, out_mode, act_slope, ptr(out), ld_out, ptr(st.alpha),
         *((ptr(nxt[0].Wp) + 4 * nxt[0].D * nxt[0].Kp, nxt[0].Kp, ptr(nxt[0].mask), feat_p, ptr(nxt[1])) if nxt is not None
           else ((None, out_drop.Kp, ptr(out_drop.mask), feat_p, None) if out_drop is not None else (None, 0, None, 0.0, None))),
         4 if _NO_EGO_WALK else (_FWD_SWEEP if H == 4 and D % 4 == 0 else 0), s)


def _gat_aggregate_bwd(csr, st, attn_p, attn_slope, d_pre, ld_dpre):
    """message/reduce backward of one layer: d_Y [N, Fp] = [d_ft | d_a1 | d_a2 | 0] from the gradient of its aggregated output"""
    N = st.X.shape[0]
    H, D, Fp = st.H, st.D, st.Fp
    F, Fe = H * D, H * D + 2 * H
    d_Y = _empty((N, Fp), st.X)
    dz = _empty((max(csr.n_edges, 1) * H,), st.X)
    call("txe_gat_aggregate_bwd", ptr(csr.rowptr_in), ptr(csr.col_src), ptr(csr.rowptr_out), ptr(csr.col_dst), ptr(csr.pos_out),
         N, ptr(st.Y), Fp, ptr(st.Y) + 4 * F, ptr(st.Y) + 4 * (F + H), Fp, H, D, attn_slope, attn_p, st.seed + 1, ptr(st.alpha),
         ptr(d_pre), ld_dpre, ptr(d_Y), Fp, ptr(d_Y) + 4 * F, ptr(d_Y) + 4 * (F + H), Fp, ptr(dz), Fp - Fe, _lib.stream_ptr())   # clears d_Y's padding too
    return d_Y


class _TailChain:
    """the deferred phase-B reductions of a stack's backward pass (include/txe.h: txe_gat_dense_bwd `chain`): host memory the C entry
    points fill, plus the workspaces / operands the deferred jobs read -- kept alive until the launch that runs them has been enqueued
    (a buffer released earlier could be handed to a later allocation of the same stream and overwritten before that launch)"""

    def __init__(self):
        import ctypes
        self.buf = ctypes.create_string_buffer(_lib.TAIL_CHAIN_BYTES)
        self.ptr = ctypes.cast(self.buf, ctypes.c_void_p)
        self.keep = []


def _gat_dense_bwd(st, pos, vocab, feat_p, d_Y, need_dh, act_on, act_slope, chain=None, defer=False):
    """projection backward of one layer from d_Y: (d_X or None, dW, d_attn_l, d_attn_r, dP).
    chain / defer: see _TailChain (defer: this layer's last reduction launch is left to the bottom layer's)"""
    N = st.X.shape[0]
    dW, dal, dar = torch.empty_like(st.W), torch.empty_like(st.al), torch.empty_like(st.ar)
    dP = torch.empty_like(st.P) if st.P is not None else None
    d_X = _empty((N, st.Kp), st.X) if (need_dh or st.Pd > 0) else None
    wsb = pure("txe_gat_dense_ws_bytes", N, st.Kh, st.Pd, st.H, st.D, vocab)
    split_dx = bool(need_dh) and not _NO_SPLIT_GEMM
    if split_dx:                                       # room for d_Y and Wp as packed planes: d_X = d_Y Wp on the bf16 pipe
        wsb += pure("txe_gat_dense_bwd_split_ws_bytes", N, st.Kh, st.Pd, st.H, st.D)
    ws = _ws(wsb, st.X)
    def run(phases):
        call("txe_gat_dense_bwd", None if getattr(st, "vx", False) else ptr(st.X), N, st.Kh, st.Pd, ptr(pos), vocab, ptr(st.Wp), ptr(st.W), ptr(st.al), ptr(st.ar), st.H, st.D, feat_p,
             ptr(st.mask), ptr(d_Y), int(need_dh), int(act_on), act_slope if act_slope else 1.0, ptr(d_X), ptr(dW), ptr(dal), ptr(dar),
             ptr(dP), int(getattr(st, "x_dropped", False)), ptr(getattr(st, "Xt", None)), phases, chain.ptr if chain is not None else None, ptr(ws), wsb, _lib.stream_ptr())
    # (a first PGAT layer's d_X -- position columns only -- is one HBM stream over d_Y, txe_dxpos.hip; every other d_X is a GEMM)
    run(7 | (16 if split_dx else 0) | (64 if (defer and chain is not None) else 0))
    if chain is not None:
        chain.keep += [ws, d_Y, st]
    return d_X, dW, dal, dar, dP


def _gat_layer_bwd(csr, st, pos, vocab, feat_p, attn_p, attn_slope, d_pre, ld_dpre, need_dh, act_on, act_slope, chain=None, defer=False):
    d_Y = _gat_aggregate_bwd(csr, st, attn_p, attn_slope, d_pre, ld_dpre)
    return _gat_dense_bwd(st, pos, vocab, feat_p, d_Y, need_dh, act_on, act_slope, chain, defer)


def _order(first, then):
    """work submitted to stream `then` from now on starts after everything already submitted to `first` (txe_stream_order: an event
    without the system-scope fence -- two streams of one device need no L2 write-back in front of the next kernel)"""
    call("txe_stream_order", first.cuda_stream, then.cuda_stream)


_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device.index)
    if s is None:
        s = _side_streams[device.index] = torch.cuda.Stream(device=device)
    return s


def _fused_bwd_ok(csr, st, sp):
    """can the folded layer `st`'s backward run fused with the message/reduce backward of the layer below `sp`?"""
    return (not _NO_FUSED_BWD and sp.alpha is not None and sp.H * sp.D == st.Kh and st.cl is not None and csr.n_edges > 0
            and pure("txe_gat_fused_bwd_supported", st.Kh, st.Pd, sp.H, sp.D) == 1)


class FoldLink:
    """What the producer of Z (GATStackFunction, cfg.final == 'collapse_z') shares with whoever consumes Z as the folded graph vector
    hg = Z W^T: the consumer's backward leaves the main part of the output layer's weight gradient here (S slices [D, Kp], summed in
    order) and hands dZ back through autograd; the producer's backward adds the attention rows' part and returns the whole dW."""
    __slots__ = ("part", "S", "fwd", "e_part", "m", "by_k", "one_col")

    def __init__(self):
        self.part, self.S = None, 0
        # the producer's weight packing: a GAT layer's Wp [Fp][Kp] (rows < D the weight) or -- by_k -- a GCN layer's Wp [Kp128][Fop] (row k, D
        # columns; row one_col holds the bias and column one_col of Z counts as 1).  by_k: `part` comes back as [Kp][D], row one_col = d_bias
        self.by_k, self.one_col = False, -1
        # with a matcher job (folded_match_job) the producer forms T before its Z sweep, the sweep leaves <T[run(g)], keep X[u]> per node
        # (e_part) and backward's <dZ, X> sweep becomes a scaling by the matcher's score gradient (m = (ds, s, apply_exp)):
        self.fwd, self.e_part, self.m = None, None, None


def walk_plan(csr):
    """the batch's plan for the egonet-walking sweeps (txe_egonet_walk_plan): a view of the graphs like the two CSR orders, built once per
    batch -- by the first backward pass that wants it, or by the loader that built the batch -- and kept on the CSR object"""
    key = id(csr.rowptr_in)                               # (the CSR views are cached on their graph: one tensor object per batch)
    ent = _WALK_PLANS.get(key)
    if ent is not None and ent[0]() is csr.rowptr_in:
        return ent[1]
    N = csr.n_nodes
    plan = torch.empty(pure("txe_egonet_walk_plan_bytes", N) // 4, dtype=torch.int32, device=csr.rowptr_in.device)
    call("txe_egonet_walk_plan", ptr(csr.rowptr_in), ptr(csr.col_src), ptr(csr.rowptr_out), ptr(csr.col_dst), ptr(csr.pos_out),
         ptr(csr.graph_off), N, csr.n_graphs, ptr(plan), _lib.stream_ptr())
    _WALK_PLANS[key] = (weakref.ref(csr.rowptr_in, lambda _ref, key=key: _WALK_PLANS.pop(key, None)), plan)   # (the plan dies with its graph)
    return plan


_WALK_PLANS = {}       # id of a CSR's rowptr_in tensor -> (weak reference to it, the plan)


def _gat_collapse_bwd_fused(csr, st, sp, pos, rpos, pw, vocab, feat_p, attn_p, attn_slope, d_hg, act_slope, chain=None, link=None):
    """txe_gat_collapse_bwd_fused: the folded layer's parameter gradients AND the layer below's d_Y in one sweep (no d_X).
    link given: d_hg IS dZ [G, Kp] (the consumer of Z folded hg = Z W^T into its own products, FoldLink)."""
    N, G, E = st.X.shape[0], csr.n_graphs, csr.n_edges
    a12, alpha, coef, wsum, gid, Z, hg = st.cl
    edot = link is not None and link.e_part is not None and link.m is not None     # the <dZ, X> sweep was done in forward (FoldLink)
    if d_hg is None:
        if not edot:
            raise RuntimeError("folded output layer: no gradient arrived for the graph vector")
        ld = st.Kp                                  # (edot: 'dZ[g]' is the matcher's ds_g T[run(g)], read from the link -- no tensor)
    else:
        d_hg, ld = _rows(d_hg)
    dW, dal, dar = torch.empty_like(st.W), torch.empty_like(st.al), torch.empty_like(st.ar)
    dP = torch.empty_like(st.P) if st.P is not None else None
    d_pw = torch.empty_like(pw) if pw is not None else None
    Fe = sp.H * sp.D + 2 * sp.H
    d_Yp = _empty((N, sp.Fp), st.X)
    dz = _empty((max(E, 1) * sp.H,), st.X)
    v = max(vocab, pw.numel() if pw is not None else 0)
    wsb = pure("txe_gat_collapse_bwd_fused_ws_bytes", N, E, G, st.Kh, st.Pd, st.D, max(v, 8), sp.H)
    ws = _ws(wsb, st.X)
    def run(phases):
        call("txe_gat_collapse_bwd_fused", ptr(csr.rowptr_in), ptr(csr.col_src), ptr(csr.rowptr_out), ptr(csr.col_dst), ptr(csr.pos_out),
             ptr(csr.graph_off), N, E, G, ptr(st.X), st.Kh, st.Pd, ptr(pos if pos is not None else rpos), v, ptr(st.Wp), ptr(st.W),
             ptr(st.al), ptr(st.ar), st.D, feat_p, ptr(st.mask), attn_slope, attn_p, st.seed + 1, ptr(pw), ptr(a12), ptr(alpha), ptr(coef),
             ptr(wsum), ptr(gid), ptr(Z), ptr(hg), st.D, ptr(d_hg), ld, act_slope if act_slope else 1.0, ptr(sp.Y), sp.Fp, sp.H, sp.D,
             attn_slope, attn_p, sp.seed + 1, ptr(sp.alpha), ptr(d_Yp), sp.Fp, sp.Fp - Fe, ptr(dz), ptr(dW), ptr(dal), ptr(dar), ptr(dP),
             ptr(d_pw), phases | (512 if edot else 0) | (1024 if _NO_EGO_WALK else 0), ptr(link.part) if (link is not None and link.S > 0) else None,
             link.S if link is not None else 0, *((ptr(link.e_part), ptr(link.m[0]), ptr(link.m[1]), int(link.m[2]), ptr(link.fwd["T"]),
                                                  ptr(link.fwd["run_id"]), ptr(zgid)) if edot else (None, None, None, 0, None, None, None)),
             ptr(plan), chain.ptr if chain is not None else None, ptr(ws), wsb, _lib.stream_ptr())
    zgid = torch.empty(max(N, 1), dtype=torch.int32, device=st.X.device) if edot else None
    plan = walk_plan(csr) if (sp.H == 4 and not _NO_EGO_WALK and not _NO_WALK_PLAN) else None
    last = 8 | (64 if chain is not None else 0)     # (with a chain the final reductions are left to the bottom layer's launch)
    if chain is not None:
        chain.keep += [ws, d_hg, st, sp, zgid] + ([link.part, link.fwd, link.m] if link is not None else [])
    if link is not None:                            # dZ given: no product left in this layer's backward, nothing for a second stream
        if ld != st.Kp and not edot:
            raise RuntimeError("folded graph vector: dZ must have the padded row pitch")
        run(4 | 256)
        run(last | 256)
    elif _NO_SIDE_STREAM:
        run(7 | last)
    else:
        # the folded layer's weight-gradient GEMM (MFMA-bound, needs only d_hg and Z) runs on a second stream under the HBM-bound
        # sweeps: complementary resources, and nothing downstream waits for it before the final reduction
        # (| 128 on every call: the product beside other kernels takes few fat k-slices, and the workspace is laid out for them)
        main, side = torch.cuda.current_stream(), _side_stream(st.X.device)
        _order(main, side)
        with torch.cuda.stream(side):
            run(2 | 128)
        run(1 | 128)
        run(4 | 128)
        _order(side, main)
        run(last | 128)
    return d_Yp, dW, dal, dar, dP, d_pw


class GATStackFunction(torch.autograd.Function):
    """params per layer: (W [H*D, Kin], attn_l [1,H,D], attn_r [1,H,D], P [vocab, Pd] or None).
    cfg.final: 'mean' -> N x D (PGAT / GAT), 'none' -> N x H x D (GATLayer), 'collapse' -> G x D: the one-head output layer folded
    behind MeanReadout (pw None) / WeightedMeanReadout (pw = position_weights.weight, rpos = node positions)."""
    accepts_gathered_rows = True

    @staticmethod
    def forward(ctx, csr, cfg, h, pos, rpos, pw, *params):
        need = getattr(cfg, "grad_enabled", True) and any(ctx.needs_input_grad)     # (see apply_stack)
        z_only = (cfg.final == "collapse_z")        # 'collapse' that stops at Z: returns (Z [G, Kp], the output layer's packed weights)
        collapse = (cfg.final == "collapse") or z_only
        table = _use_table(h, need, cfg.feat_p) and not (collapse and cfg.n_layers == 1)
        if isinstance(h, GatheredRows) and not table:
            h = h.tensor()
        if table:
            _need_cuda(h.table, *[p for p in params if p is not None])
            src, ld_h = h, 0
        else:
            _need_cuda(h, *[p for p in params if p is not None])
            h, ld_h = _rows(h)
            src = h
        pos = _i32(pos, h.device)
        rpos = _i32(rpos, h.device) if (collapse and pw is not None) else None
        pwf = _f32(pw.reshape(-1)) if (collapse and pw is not None) else None
        L = cfg.n_layers
        N = h.shape[0]
        states = []
        with _lib.on_device(h.device):
            kh = h.shape[1]
            ref = h.table if table else h
            for l in range(L):
                st = _GatLayerState()
                st.W, st.al, st.ar, st.P = (_f32(p) for p in params[4 * l:4 * l + 4])
                st.H, st.D, st.Kh = cfg.heads[l], cfg.out_dims[l], kh
                st.Pd = 0 if st.P is None else st.P.shape[1]
                st.Kp = pure("txe_gat_padded_k", st.Kh, st.Pd)
                st.Fp = pure("txe_gat_padded_f", st.H, st.D)
                st.seed = cfg.seed + 16 * l
                st.X = None
                states.append(st)
                kh = st.H * st.D
            h = ref                                  # (allocation reference from here on; the features travel as `src`)
            states[0].X = None if table else _empty((N, states[0].Kp), h)
            if N > 0:                                # every layer's input buffer now, and ONE preparation launch for the whole stack
                for l in range(1, L):
                    states[l].X = _empty((N, states[l].Kp), h)
                # (a layer that is not the folded one: only its GEMMs read X, so X is stored with the dropout applied -- by the
                #  preparation (raw features, position columns) and by the aggregation of the layer below (_drops_output))
                states[0].vx = (not table) and _virtual_x_ok(states[0], src, ld_h, N, need, collapse and L == 1)
                _gat_layers_prepare([(st, (src if l == 0 else None), (ld_h if l == 0 else 0), pos if st.P is not None else None,
                                      _x_dropped_ok(cfg, states, l, collapse))
                                     for l, st in enumerate(states) if not (table and l == 0)], cfg.feat_p)
            fused_a12 = None
            for l, st in enumerate(states):
                last = (l == L - 1)
                F = st.H * st.D
                if last and collapse:
                    res = _gat_collapse_fwd(csr, st, src if l == 0 else None, ld_h if l == 0 else 0, pos if st.P is not None else None,
                                            rpos, pwf, cfg.feat_p, cfg.attn_p, cfg.attn_slope, a12=fused_a12, z_only=z_only,
                                            fold_job=getattr(cfg, "fold_job", None) if (z_only and need) else None,
                                            link=getattr(cfg, "link", None))
                    if z_only:
                        res = (res, st.Wp)
                        ctx.mark_non_differentiable(st.Wp)
                        ctx.set_materialize_grads(False)    # (no zero "gradient" of the packed weights: a 4 MB fill per step)
                    if not need:
                        st.cl = st.mask = st.Wp = st.X = None
                    break
                if last:
                    out, ld_out = _empty((N, F), h), F
                else:                                  # the aggregation writes straight into the next layer's padded input
                    if states[l + 1].X is None:
                        states[l + 1].X = _empty((N, states[l + 1].Kp), h)
                    out, ld_out = states[l + 1].X, states[l + 1].Kp
                out_mode = 0 if (last or cfg.act_slope is None) else 1
                nxt = None
                if (collapse and l + 1 == L - 1 and N > 0 and st.D % 4 == 0 and states[l + 1].Kp - st.H * st.D <= 128 and states[l + 1].Kp <= 4096
                        and not _NO_FUSED_LOGITS):
                    # the folded output layer is prepared first: its keep mask and folded attention rows feed this layer's epilogue
                    sn = states[l + 1]
                    _gat_layer_prepare(sn, None, 0, pos if sn.P is not None else None, cfg.feat_p)
                    fused_a12 = _empty((N, 2), h)
                    nxt = (sn, fused_a12)
                # (the layer above reads its input through plain GEMM operands: this layer's aggregation applies that layer's dropout)
                out_drop = states[l + 1] if (not last and nxt is None and getattr(states[l + 1], "x_dropped", False)) else None
                _gat_layer_fwd(csr, st, src if l == 0 else None, ld_h if l == 0 else 0, pos if st.P is not None else None, out, ld_out,
                               cfg.feat_p, cfg.attn_p, cfg.attn_slope, out_mode, cfg.act_slope or 1.0, need, nxt, out_drop)
                if not need:
                    st.Y = st.mask = st.Wp = None
                    if l > 0:
                        st.X = None
            H, D = cfg.heads[-1], cfg.out_dims[-1]
            if collapse:
                pass
            elif cfg.final == "mean":
                if H == 1:
                    res = out.view(N, D)
                else:
                    res = _empty((N, D), h)
                    call("txe_head_mean_fwd", ptr(out), H, D, N, ptr(res), _lib.stream_ptr())
            else:
                res = out.view(N, H, D)
        ctx.csr, ctx.cfg, ctx.pos, ctx.states = csr, cfg, pos, (states if need else None)
        ctx.rpos, ctx.pwf, ctx.pw_shape = rpos, pwf, (pw.shape if pwf is not None else None)
        ctx.h_req = ctx.needs_input_grad[2]
        ctx.param_ids, ctx.pw_id = [id(p) for p in params], id(pw)
        ctx.link = getattr(cfg, "link", None) if z_only else None
        note_route("stack", cfg.final + ("+edot" if (z_only and ctx.link is not None and ctx.link.e_part is not None) else ""))
        if _CAPTURE is not None:
            _CAPTURE.append((csr, cfg, states))
        return res

    @staticmethod
    def backward(ctx, d_res, *_unused):
        csr, cfg, pos, states = ctx.csr, ctx.cfg, ctx.pos, ctx.states
        if states is None:
            raise RuntimeError(_BACKWARD_TWICE)
        L = cfg.n_layers
        H, D = cfg.heads[-1], cfg.out_dims[-1]
        z_only = (cfg.final == "collapse_z")
        collapse = (cfg.final == "collapse") or z_only
        link = ctx.link if z_only else None
        edot = link is not None and link.e_part is not None and link.m is not None
        if d_res is None and not edot:
            # (collapse_z does not materialise absent gradients: Z took no part in the loss -- nothing to propagate, and the saved
            #  state can go)
            ctx.states = None
            return (None,) * (6 + 4 * L)
        d_res = None if edot else _f32(d_res)       # (edot: the matcher's 'dZ' travels through the FoldLink, not through autograd)
        N = states[0].X.shape[0]
        grads = [None] * (4 * L)
        d_pw = None
        note_route("stack_bwd", "fused+edot" if edot else ("collapse" if collapse else "layers"))
        with _lib.on_device(states[0].X.device):
            if collapse:
                d_pre, ld_dpre = None, 0
            elif cfg.final == "mean" and H > 1:
                d_pre = _empty((N, H * D), d_res)
                call("txe_head_mean_bwd", ptr(d_res), H, D, N, ptr(d_pre), _lib.stream_ptr())
            else:
                d_pre = d_res.reshape(N, H * D)
            if d_pre is not None:
                ld_dpre = d_pre.stride(0)
            d_X = None
            d_Y_ready = None                       # d_Y of layer l already produced by the fused sweep of layer l+1
            # the layers' last reduction launches (parameter gradients only) are chained into the bottom layer's -- unless somebody
            # wants every layer's gradients the moment its backward ends (the overlapped gradient all-reduce)
            chain = _TailChain() if (_GRAD_READY is None and L > 1 and not _NO_TAIL_CHAIN) else None
            for l in range(L - 1, -1, -1):
                st = states[l]
                need_dh = (l > 0) or ctx.h_req
                # the input of layer l>0 is leaky_relu(out_{l-1}) (fused epilogue): fold its derivative into dX
                act_on = (l > 0 and cfg.act_slope is not None)
                if collapse and l == L - 1:
                    if l > 0 and _fused_bwd_ok(csr, st, states[l - 1]):
                        d_Y_ready, dW, dal, dar, dP, d_pw = _gat_collapse_bwd_fused(
                            csr, st, states[l - 1], pos if st.P is not None else None, ctx.rpos, ctx.pwf, cfg.vocab, cfg.feat_p, cfg.attn_p,
                            cfg.attn_slope, d_res, cfg.act_slope if act_on else None, chain, link=(ctx.link or FoldLink()) if z_only else None)
                    elif z_only:
                        raise RuntimeError("collapse_z was requested for a stack whose fused backward does not apply (folded_graph_vector_ok)")
                    else:
                        d_X, dW, dal, dar, dP, d_pw = _gat_collapse_bwd(csr, st, pos if st.P is not None else None, ctx.rpos, ctx.pwf, cfg.vocab,
                                                                        cfg.feat_p, cfg.attn_p, cfg.attn_slope, d_res, act_on, cfg.act_slope)
                elif d_Y_ready is not None:
                    d_X, dW, dal, dar, dP = _gat_dense_bwd(st, pos if st.P is not None else None, cfg.vocab, cfg.feat_p, d_Y_ready, need_dh,
                                                           act_on, cfg.act_slope, chain, defer=l > 0)
                    d_Y_ready = None
                else:
                    d_X, dW, dal, dar, dP = _gat_layer_bwd(csr, st, pos if st.P is not None else None, cfg.vocab, cfg.feat_p, cfg.attn_p,
                                                           cfg.attn_slope, d_pre, ld_dpre, need_dh, act_on, cfg.act_slope, chain, defer=l > 0)
                grads[4 * l:4 * l + 4] = [dW, dal, dar, dP]
                if _GRAD_READY is not None:               # (tensors, ids of the parameters they are the gradients of)
                    last_c = collapse and l == L - 1
                    _GRAD_READY(l, [dW, dal, dar, dP] + ([d_pw] if last_c else []), ctx.param_ids[4 * l:4 * l + 4] + ([ctx.pw_id] if last_c else []))
                if l > 0 and d_Y_ready is None:
                    d_pre, ld_dpre = d_X, st.Kp            # its first H*D(l-1) columns are d(pre-activation out_{l-1})
            d_h = d_X[:, :states[0].Kh].contiguous() if ctx.h_req else None
            if chain is not None:                  # (nothing left unless the bottom layer took a route without a phase B of its own)
                call("txe_gat_tail_flush", chain.ptr, _lib.stream_ptr())
                chain.keep = []
            if _GRAD_FLUSH is not None:
                _GRAD_FLUSH()
        ctx.states = None
        if d_pw is not None:
            d_pw = d_pw.reshape(ctx.pw_shape)
        return (None, None, d_h, None, None, d_pw, *grads)


# ================================================================================================================
# GCN stack (PGCN / GCN / a single GCNLayer)
# ================================================================================================================
class GCNConfig:
    def __init__(self, out_dims, vocab, act_slopes, drop_ps, seed):
        self.out_dims, self.vocab = list(out_dims), vocab
        self.act_slopes = list(act_slopes)      # per layer: slope of the fused leaky_relu or None
        self.drop_ps = [float(p) for p in drop_ps]
        self.seed = int(seed)
        self.n_layers = len(self.out_dims)


class _GcnLayerState:
    __slots__ = ("X", "Wp", "mask", "W", "b", "P", "Kh", "Pd", "Kp", "Fo", "Fop", "seed", "cl", "x_dropped")


class GCNStackFunction(torch.autograd.Function):
    """params per layer: (W [Kin, Fo], bias [Fo] or None, P [vocab, Pd] or None).
    cfg.final == 'collapse': G x Fo -- the (activation-free) output layer folded behind MeanReadout (pw None) /
    WeightedMeanReadout (txe_gcn_collapse_*)."""

    accepts_gathered_rows = True

    @staticmethod
    def forward(ctx, csr, cfg, h, pos, rpos, pw, *params):
        z_only = (getattr(cfg, "final", None) == "collapse_z")   # 'collapse' that stops at Z: returns (Z [G, Kp], the output layer's packed weights)
        collapse = (getattr(cfg, "final", None) == "collapse") or z_only
        L = cfg.n_layers
        need = getattr(cfg, "grad_enabled", True) and any(ctx.needs_input_grad)     # (see apply_stack)
        table = _use_table(h, need, cfg.drop_ps[0]) and not (collapse and L == 1)
        if isinstance(h, GatheredRows) and not table:
            h = h.tensor()
        src = h if table else None                 # features as rows of a table: layer 0 projects the table
        if table:
            _need_cuda(h.table, *[p for p in params if p is not None])
            N, kh0, ld_h, h = h.index.shape[0], h.table.shape[1], 0, h.table       # h: allocation reference from here on
        else:
            _need_cuda(h, *[p for p in params if p is not None])
            h, ld_h = _rows(h)
            N, kh0 = h.shape
        pos = _i32(pos, h.device)
        rpos = _i32(rpos, h.device) if (collapse and pw is not None) else None
        pwf = _f32(pw.reshape(-1)) if (collapse and pw is not None) else None
        states = []
        with _lib.on_device(h.device):
            st_ = _lib.stream_ptr()
            kh = kh0
            for l in range(L):
                st = _GcnLayerState()
                st.W, st.b, st.P = (_f32(p) for p in params[3 * l:3 * l + 3])
                st.Kh, st.Fo = kh, cfg.out_dims[l]
                st.Pd = 0 if st.P is None else st.P.shape[1]
                st.Kp = pure("txe_gat_padded_k", st.Kh, st.Pd)
                st.Fop = pure("txe_gcn_padded_f", st.Fo)
                st.seed = cfg.seed + 16 * l
                st.X = None
                states.append(st)
                kh = st.Fo
            # every layer's input buffer now; ONE launch prepares the whole stack (layer inputs' position / padding columns, packed weights,
            # keep masks -- none of it depends on a layer below's output) and forms the degree normalisation
            norm = _empty((max(N, 1),), h)
            for l, st in enumerate(states):
                st.X = None if (table and l == 0) else _empty((N, st.Kp), h)
            todo = [(l, st) for l, st in enumerate(states) if not (table and l == 0)]
            import ctypes
            descs = (_lib.GcnPrepareDesc * max(len(todo), 1))()
            keep = []
            for d, (l, st) in zip(descs, todo):
                last = (l == L - 1)
                kp128 = (st.Kp + 127) // 128 * 128
                st.Wp = _empty((kp128, st.Fop), h)
                st.mask = (torch.empty((N, (st.Kh + st.Pd + 31) // 32), dtype=torch.int32, device=h.device)
                           if cfg.drop_ps[l] > 0.0 else None)
                # (a first layer on raw features that is not the folded one: only its GEMMs read X -> stored with the dropout applied)
                st.x_dropped = bool(l == 0 and not (last and collapse) and cfg.drop_ps[l] > 0.0)
                # (folded into the matcher: the bias rides as one more weight row, behind a column of Z that counts as 1)
                bias_row = st.b if (last and z_only and st.b is not None) else None
                keep.append(bias_row)
                d.h, d.ld_h, d.n_nodes, d.Kh = ptr(h if l == 0 else None), (ld_h if l == 0 else 0), N, st.Kh
                d.pos, d.P, d.Pd, d.X = ptr(pos if st.P is not None else None), ptr(st.P), st.Pd, ptr(st.X)
                d.W, d.Fo, d.Wp, d.drop_p, d.seed, d.mask = ptr(st.W), st.Fo, ptr(st.Wp), cfg.drop_ps[l], st.seed, ptr(st.mask)
                d.x_dropped, d.bias_row = int(st.x_dropped), ptr(bias_row)
            if todo:
                call("txe_gcn_layers_prepare", ctypes.cast(descs, ctypes.c_void_p), len(todo), ptr(csr.rowptr_in), csr.n_nodes, ptr(norm), st_)
            else:
                call("txe_gcn_norm", ptr(csr.rowptr_in), csr.n_nodes, ptr(norm), st_)
            tws = _tail_ws(h)
            for l, st in enumerate(states):
                last = (l == L - 1)
                if last and collapse:
                    G = csr.n_graphs
                    coef, wsum = _empty((max(N, 1),), h), _empty((max(G, 1),), h)
                    gid = torch.empty(max(N, 1), dtype=torch.int32, device=h.device)
                    Z, out = _empty((max(G, 1), st.Kp), h), (None if z_only else _empty((G, st.Fo), h))
                    wsb = pure("txe_gcn_collapse_ws_bytes", N, G, st.Kh, st.Pd, st.Fo, 8)
                    ws = _ws(wsb, h)
                    call("txe_gcn_collapse_fwd", ptr(csr.rowptr_out), ptr(csr.col_dst), ptr(csr.graph_off), N, G, ptr(st.X), st.Kh, st.Pd,
                         ptr(st.Wp), st.Fo, ptr(st.b), cfg.drop_ps[l], ptr(st.mask), ptr(norm), ptr(rpos), ptr(pwf), ptr(coef), ptr(wsum),
                         ptr(gid), ptr(Z), ptr(out), st.Fo, ptr(ws), wsb, st_)
                    st.cl = (coef, wsum, gid, Z)
                    if z_only:
                        link = getattr(cfg, "link", None)
                        if link is not None:
                            link.by_k, link.one_col = True, (st.Kh + st.Pd if st.b is not None else -1)
                        out = (Z, st.Wp)
                        ctx.mark_non_differentiable(st.Wp)
                        ctx.set_materialize_grads(False)
                    if not need:
                        st.cl = st.mask = st.Wp = st.X = None
                    break
                hw = _empty((N, st.Fop), h)
                if table and l == 0:
                    T, T2 = _gcn_table_projection(st, src)[:2]
                    call("txe_gather_add_rows", ptr(T), st.Fop, ptr(_i32(src.index, h.device)), ptr(T2), st.Fop,
                         ptr(pos) if T2 is not None else None, N, st.Fop, ptr(hw), st.Fop, st_)
                    st.mask = st.Wp = None
                else:
                    dropped = getattr(st, "x_dropped", False)
                    call("txe_gcn_dense_fwd", ptr(st.X), N, st.Kh, st.Pd, ptr(st.Wp), st.Fo, 0.0 if dropped else cfg.drop_ps[l],
                         None if dropped else ptr(st.mask), ptr(hw), ptr(tws), tws.numel(), st_)
                    _launch_pending_prefetch()
                if last:
                    out, ld_out = _empty((N, st.Fo), h), st.Fo
                else:
                    out, ld_out = states[l + 1].X, states[l + 1].Kp
                slope = cfg.act_slopes[l]
                call("txe_gcn_aggregate_fwd", ptr(csr.rowptr_in), ptr(csr.col_src), N, ptr(hw), st.Fop, ptr(norm), ptr(st.b),
                     0 if slope is None else 1, slope or 1.0, st.Fo, ptr(out), ld_out, st_)
                if not need:
                    st.mask = st.Wp = None
                    if l > 0:
                        st.X = None
        ctx.csr, ctx.cfg, ctx.pos, ctx.norm = csr, cfg, pos, norm
        ctx.link = getattr(cfg, "link", None) if z_only else None
        note_route("stack", "collapse_z" if z_only else ("collapse" if collapse else "layers"))
        if _CAPTURE is not None:
            _CAPTURE.append((csr, cfg, states))
        ctx.states = states if need else None
        ctx.h_req = ctx.needs_input_grad[2]
        ctx.out = out if (need and not z_only) else None
        ctx.rpos, ctx.pwf, ctx.pw_shape = rpos, pwf, (pw.shape if pwf is not None else None)
        return out

    @staticmethod
    def backward(ctx, d_out, *_unused):
        csr, cfg, pos, norm, states = ctx.csr, ctx.cfg, ctx.pos, ctx.norm, ctx.states
        if states is None:
            raise RuntimeError(_BACKWARD_TWICE)
        L = cfg.n_layers
        z_only = (getattr(cfg, "final", None) == "collapse_z")
        if d_out is None:                          # (collapse_z does not materialise absent gradients: Z took no part in the loss)
            ctx.states = None
            return (None,) * (6 + 3 * L)
        d_out = _f32(d_out)
        grads = [None] * (3 * L)
        collapse = (getattr(cfg, "final", None) == "collapse") or z_only
        d_pw = None
        with _lib.on_device(d_out.device):
            st_ = _lib.stream_ptr()
            N = states[0].X.shape[0]
            if collapse:
                d_pre = None
            elif cfg.act_slopes[-1] is not None:   # a standalone activated layer: undo the fused activation explicitly
                d_pre = torch.empty_like(d_out)
                call("txe_leaky_relu_bwd", ptr(d_out), ptr(ctx.out), cfg.act_slopes[-1], d_out.numel(), ptr(d_pre), st_)
            else:
                d_pre = d_out
            ld_dpre = d_pre.stride(0) if d_pre is not None else 0
            d_X = None
            for l in range(L - 1, -1, -1):
                st = states[l]
                if collapse and l == L - 1:
                    coef, wsum, gid, Z = st.cl
                    G = csr.n_graphs
                    dh, ld = _rows(d_out)
                    act_on = l > 0 and cfg.act_slopes[l - 1] is not None
                    d_X = _empty((N, st.Kp), d_out)
                    if z_only:                     # d_out IS dZ; the consumer of Z left dW (and d_b as row Kt) in the FoldLink, [Kp][Fo]
                        part = ctx.link.part if ctx.link is not None else None
                        if part is None or ctx.link.S != 1 or tuple(part.shape) != (st.Kp, st.Fo):
                            raise RuntimeError("folded GCN output layer: the consumer of Z left no weight gradient in the FoldLink")
                        Kt = st.Kh + st.Pd
                        dW, d_b = part[:Kt], (part[Kt] if st.b is not None else None)
                    else:
                        dW = torch.empty_like(st.W)
                        d_b = torch.empty_like(st.b) if st.b is not None else None
                    dP = torch.empty_like(st.P) if st.P is not None else None
                    d_pw = torch.empty_like(ctx.pwf) if ctx.pwf is not None else None
                    v = max(cfg.vocab, ctx.pwf.numel() if ctx.pwf is not None else 0)
                    wsb = pure("txe_gcn_collapse_ws_bytes", N, G, st.Kh, st.Pd, st.Fo, max(v, 8))
                    ws = _ws(wsb, d_out)
                    call("txe_gcn_collapse_bwd", ptr(csr.rowptr_in), ptr(csr.col_src), ptr(csr.graph_off), N, G, ptr(st.X), st.Kh, st.Pd,
                         ptr(pos if st.P is not None else ctx.rpos), v, ptr(st.Wp), st.Fo, cfg.drop_ps[l], ptr(st.mask), ptr(norm),
                         ptr(ctx.pwf), ptr(coef), ptr(wsum), ptr(gid), ptr(Z), ptr(dh), ld, int(act_on),
                         (cfg.act_slopes[l - 1] if act_on else 1.0), ptr(d_X), None if z_only else ptr(dW), None if z_only else ptr(d_b),
                         ptr(dP), ptr(d_pw), int(z_only), ptr(ws), wsb, st_)
                    grads[3 * l:3 * l + 3] = [dW, d_b, dP]
                    if l > 0:
                        d_pre, ld_dpre = d_X, st.Kp
                    continue
                d_hw = _empty((N, st.Fop), d_out)
                d_b = torch.empty_like(st.b) if st.b is not None else None
                wsb = pure("txe_gcn_aggregate_bwd_ws_bytes", N, st.Fo)
                ws = _ws(wsb, d_out)
                call("txe_gcn_aggregate_bwd", ptr(csr.rowptr_out), ptr(csr.col_dst), N, ptr(d_pre), ld_dpre, ptr(norm), st.Fo, ptr(d_hw),
                     st.Fop, ptr(d_b), ptr(ws), wsb, st_)
                need_dh = (l > 0) or ctx.h_req
                act_on = l > 0 and cfg.act_slopes[l - 1] is not None
                d_X = _empty((N, st.Kp), d_out) if (need_dh or st.Pd > 0) else None
                dW = torch.empty_like(st.W)
                dP = torch.empty_like(st.P) if st.P is not None else None
                wsb2 = pure("txe_gcn_dense_ws_bytes", N, st.Kh, st.Pd, st.Fo, cfg.vocab)
                ws2 = _ws(wsb2, d_out)
                call("txe_gcn_dense_bwd", ptr(st.X), N, st.Kh, st.Pd, ptr(pos if st.P is not None else None), cfg.vocab, ptr(st.Wp), st.Fo,
                     cfg.drop_ps[l], ptr(st.mask), ptr(d_hw), int(need_dh), int(act_on), (cfg.act_slopes[l - 1] if act_on else 1.0),
                     ptr(d_X), ptr(dW), ptr(dP), int(getattr(st, "x_dropped", False)), ptr(ws2), wsb2, st_)
                grads[3 * l:3 * l + 3] = [dW, d_b, dP]
                if l > 0:
                    d_pre, ld_dpre = d_X, st.Kp
            d_h = d_X[:, :states[0].Kh].contiguous() if ctx.h_req else None
        ctx.states = None
        if d_pw is not None:
            d_pw = d_pw.reshape(ctx.pw_shape)
        return (None, None, d_h, None, None, d_pw, *grads)


# ================================================================================================================
# Readout (MeanReadout / WeightedMeanReadout)
# ================================================================================================================
class ReadoutFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, csr, h, pos, pw):
        _need_cuda(h, pw)
        h, ld_h = _rows(h)
        G, D = csr.n_graphs, h.shape[1]
        pos = _i32(pos, h.device) if pw is not None else None
        pwf = _f32(pw.reshape(-1)) if pw is not None else None
        hg, wsum = _empty((G, D), h), _empty((max(G, 1),), h)
        with _lib.on_device(h.device):
            call("txe_readout_fwd", ptr(csr.graph_off), G, ptr(h), ld_h, ptr(pos), ptr(pwf), D, ptr(hg), ptr(wsum),
                 _lib.stream_ptr())
        ctx.csr, ctx.pos, ctx.misc = csr, pos, (h, ld_h, pwf, hg, wsum)
        ctx.pw_shape = None if pw is None else pw.shape
        return hg

    @staticmethod
    def backward(ctx, d_hg):
        csr, pos = ctx.csr, ctx.pos
        h, ld_h, pwf, hg, wsum = ctx.misc
        G, D = csr.n_graphs, h.shape[1]
        d_hg = _f32(d_hg)
        d_h = _empty((h.shape[0], D), h)
        vocab = 0 if pwf is None else pwf.numel()
        d_pw = torch.empty_like(pwf) if pwf is not None else None
        ws = _empty((max(G, 1) * max(vocab, 1),), h) if pwf is not None else None
        with _lib.on_device(h.device):
            call("txe_readout_bwd", ptr(csr.graph_off), G, ptr(h), ld_h, ptr(pos), ptr(pwf), vocab, D, ptr(hg), ptr(wsum), ptr(d_hg),
                 ptr(d_h), D, ptr(d_pw), ptr(ws), _lib.stream_ptr())
        return None, d_h, None, (d_pw.reshape(ctx.pw_shape) if d_pw is not None else None)


class ReadoutMultiFunction(torch.autograd.Function):
    """mode 1 SumReadout, 2 MaxReadout, 3 ConcatReadout (model_zoo.py:244-276)"""

    @staticmethod
    def forward(ctx, csr, h, pos, mode):
        _need_cuda(h)
        h, ld_h = _rows(h)
        G, D = csr.n_graphs, h.shape[1]
        pos = _i32(pos, h.device) if mode == 3 else None
        hg = _empty((G, 3 * D if mode == 3 else D), h)
        argmax = torch.empty((max(G, 1), D), dtype=torch.int32, device=h.device) if mode == 2 else None
        with _lib.on_device(h.device):
            call("txe_readout_multi_fwd", ptr(csr.graph_off), G, ptr(h), ld_h, ptr(pos), D, mode, ptr(hg), ptr(argmax), _lib.stream_ptr())
        ctx.misc = (csr, pos, mode, argmax, h.shape[0], D)
        return hg

    @staticmethod
    def backward(ctx, d_hg):
        csr, pos, mode, argmax, N, D = ctx.misc
        d_hg = _f32(d_hg)
        d_h = _empty((N, D), d_hg)
        with _lib.on_device(d_hg.device):
            call("txe_readout_multi_bwd", ptr(csr.graph_off), csr.n_graphs, ptr(pos), D, mode, ptr(d_hg), ptr(argmax), ptr(d_h), D,
                 _lib.stream_ptr())
        return None, d_h, None, None


class LinearFunction(torch.autograd.Function):
    """y = act([x1 | x2] W^T + b): nn.Linear over a virtual concat (the MLP matcher, model_zoo.py:285-298); act 0/1 relu/2 tanh"""

    @staticmethod
    def forward(ctx, x1, x2, W, b, act):
        _need_cuda(x1, x2, W, b)
        x1, ld1 = _rows(x1)
        l = x1.shape[1]
        if x2 is not None:
            x2, ld2 = _rows(x2)
            r = x2.shape[1]
        else:
            ld2, r = 0, 0
        Wf, bf = _f32(W), _f32(b)
        G, O = x1.shape[0], Wf.shape[0]
        y = _empty((G, O), x1)
        with _lib.on_device(x1.device):
            call("txe_linear_fwd", ptr(x1), ld1, l, ptr(x2), ld2, r, G, ptr(Wf), ptr(bf), O, int(act), ptr(y), _lib.stream_ptr())
        ctx.misc = (x1, ld1, l, x2, ld2, r, Wf, bf is not None, int(act), y)
        ctx.req = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, ld1, l, x2, ld2, r, Wf, has_b, act, y = ctx.misc
        dy = _f32(dy)
        G, O = y.shape
        need1, need2 = ctx.req
        dx1 = _empty((G, l), y) if (need1 or need2) else None
        dx2 = _empty((G, r), y) if (need2 and x2 is not None) else None
        dW = torch.empty_like(Wf)
        db = _empty((O,), y) if has_b else None
        with _lib.on_device(y.device):
            wsb = pure("txe_linear_bwd_ws_bytes", G, l, r, O)
            ws = _ws(wsb, y)
            call("txe_linear_bwd", ptr(x1), ld1, l, ptr(x2), ld2, r, G, ptr(Wf), O, act, ptr(y), ptr(dy), ptr(dx1), l, ptr(dx2), r, ptr(dW),
                 ptr(db), ptr(ws), wsb, _lib.stream_ptr())
        return (dx1 if need1 else None), (dx2 if need2 else None), dW, db, None


# ================================================================================================================
# Bilinear match (BIM / LBM) -- pairwise form of training, model.py:86
# ================================================================================================================
def bilinear_query_prefetch(e2, W):
    """V = e2 W^T of the query-side match (BilinearPairFunction), launched on the second stream: it depends on the queries and the
    matcher's weight only, so it can run under the encoder (TaxoExpan.forward calls this before graph_propagate).  Returns a token for
    BilinearPairFunction.apply(..., pre=token); None when there is nothing to gain (gradient wanted for e2, CPU tensors, no side stream)."""
    if _NO_SIDE_STREAM or not (torch.is_tensor(e2) and e2.is_cuda and W.is_cuda) or e2.requires_grad or e2.dim() != 2 or e2.shape[0] == 0:
        return None
    e2c, ld2 = _rows(e2)
    Wf = _f32(W).reshape(W.shape[-2], W.shape[-1])
    G, r = e2c.shape
    l = Wf.shape[0]
    main, side = torch.cuda.current_stream(e2.device), _side_stream(e2.device)
    V = _empty((G, l), e2c)
    tok = dict(V=V, e2=e2, e2_version=e2._version, W=W, W_version=W._version, stream=side, e2c=e2c, launched=False)

    def launch(on_side=True):
        if tok["launched"]:
            return
        tok["launched"] = True
        if on_side:
            _order(torch.cuda.current_stream(e2.device), side)
        with _lib.on_device(e2.device), torch.cuda.stream(side if on_side else torch.cuda.current_stream(e2.device)):
            call("txe_bilinear_query_project", ptr(e2c), ld2, G, l, r, ptr(Wf), ptr(V), _lib.stream_ptr())
        if on_side:
            # V was allocated on the caller's stream and is written on the second one: tell the caching allocator, so that a token
            # that is never consumed (rejected by forward, an exception in between) cannot hand V's block to a main-stream tensor
            # while the projection is still writing it; the same for the operands it reads
            for t in (V, e2c, Wf):
                t.record_stream(side)
        tok["stream"] = side if on_side else None
    tok["launch"] = launch
    # launched by the encoder behind its first projection GEMM (_launch_pending_prefetch): started at the very beginning its workgroups
    # take slots before the persistent first-layer projection's, whose late starters then finish late
    del _pending_prefetch[:]                    # (a token nobody launched holds no device work: dropping it is safe)
    _pending_prefetch.append(tok)
    return tok


_pending_prefetch = []


def _launch_pending_prefetch():
    while _pending_prefetch:
        _pending_prefetch.pop()["launch"]()


class BilinearPairFunction(torch.autograd.Function):
    """s_i = e1_i^T W e2_i (exp optionally).  When e2 needs no gradient (queries: always so in training) the query-side form runs:
    V = e2 W^T in forward makes backward's d_e1 = dsl * V elementwise (txe_bilinear_query_*); otherwise the candidate-side form
    U = e1 W with gradients to both inputs (txe_bilinear_pair_*).  pre: a bilinear_query_prefetch token whose V is used if it still
    matches e2 / W."""

    @staticmethod
    def forward(ctx, e1, e2, W, apply_exp, pre=None):
        _need_cuda(e1, e2, W)
        e1, ld1 = _rows(e1)
        e2_in = e2
        e2, ld2 = _rows(e2)
        Wf = _f32(W).reshape(W.shape[-2], W.shape[-1])
        G, l = e1.shape
        r = e2.shape[1]
        s = _empty((G,), e1)
        query_side = not ctx.needs_input_grad[1]
        with _lib.on_device(e1.device):
            if query_side:
                ready = (pre is not None and pre["e2"] is e2_in and pre["e2_version"] == e2_in._version and pre["W"] is W
                         and pre["W_version"] == W._version and tuple(pre["V"].shape) == (G, l))
                if ready:
                    U = pre["V"]
                    pre["launch"](on_side=False)                 # (nobody started it: in line, on this stream)
                    if pre["stream"] is not None:
                        _order(pre["stream"], torch.cuda.current_stream())
                    call("txe_bilinear_query_dot", ptr(e1), ld1, ptr(U), G, l, int(apply_exp), ptr(s), _lib.stream_ptr())
                else:
                    U = _empty((max(G, 1), l), e1)          # V = e2 W^T
                    call("txe_bilinear_query_fwd", ptr(e1), ld1, ptr(e2), ld2, G, l, r, ptr(Wf), int(apply_exp), ptr(U), ptr(s), _lib.stream_ptr())
            else:
                U = _empty((max(G, 1), r), e1)
                call("txe_bilinear_pair_fwd", ptr(e1), ld1, ptr(e2), ld2, G, l, r, ptr(Wf), int(apply_exp), ptr(U), ptr(s),
                     _lib.stream_ptr())
        ctx.misc = (e1, ld1, e2, ld2, Wf, U, s, int(apply_exp), W.shape, query_side)
        return s.unsqueeze(1)

    @staticmethod
    def backward(ctx, ds):
        e1, ld1, e2, ld2, Wf, U, s, apply_exp, wshape, query_side = ctx.misc
        G, l = e1.shape
        r = e2.shape[1]
        ds = _f32(ds.reshape(-1))
        d_e1 = _empty((G, l), e1)
        dW = torch.empty_like(Wf)
        with _lib.on_device(e1.device):
            if query_side:
                wsb = pure("txe_bilinear_query_bwd_ws_bytes", G, l, r)
                ws = _ws(wsb, e1)
                call("txe_bilinear_query_bwd", ptr(e1), ld1, ptr(e2), ld2, G, l, r, apply_exp, ptr(U), ptr(s), ptr(ds), ptr(d_e1), l, ptr(dW),
                     ptr(ws), wsb, _lib.stream_ptr())
                return d_e1, None, dW.reshape(wshape), None, None
            d_e2 = _empty((G, r), e1)
            wsb = pure("txe_bilinear_pair_bwd_ws_bytes", G, l, r)
            ws = _ws(wsb, e1)
            call("txe_bilinear_pair_bwd", ptr(e1), ld1, ptr(e2), ld2, G, l, r, ptr(Wf), apply_exp, ptr(U), ptr(s), ptr(ds), ptr(d_e1),
                 l, ptr(d_e2), r, ptr(dW), ptr(ws), wsb, _lib.stream_ptr())
        return d_e1, d_e2, dW.reshape(wshape), None, None


class RepeatedRows:
    """A [G, r] matrix whose rows repeat in RUNS, kept as its U distinct rows: the query features of a training batch -- one query is
    paired with 1 + negative_size consecutive anchors and the reference's collate stacks its row once per pair (data_loaders.py:9-28).
    rows [U, r] (device), run_off [U + 1] int32 (device; first pair of every run, run_off[U] = G).  BIM / LBM take it as their query
    argument and project U rows instead of G (BilinearRunsFunction); every other consumer calls dense().  data_loaders.DeviceBatchLoader
    yields it as the query features (repeated_queries=True)."""

    def __init__(self, rows, run_off, n_rows):
        self.rows, self.run_off, self.n_rows = rows, run_off, int(n_rows)
        assert run_off.dtype == torch.int32 and run_off.numel() == rows.shape[0] + 1

    @staticmethod
    def from_ids(table, ids, device=None):
        """table[ids] with the runs of equal consecutive ids found on the host (ids: host int array)"""
        import numpy as np
        ids = np.asarray(ids).reshape(-1)
        start = np.flatnonzero(np.concatenate([[True], ids[1:] != ids[:-1]])) if ids.size else np.zeros(0, dtype=np.int64)
        dev = table.device if device is None else device
        off = torch.from_numpy(np.concatenate([start, [ids.size]]).astype(np.int32)).to(dev)
        return RepeatedRows(table.index_select(0, torch.from_numpy(ids[start].astype(np.int64)).to(table.device)).to(dev), off, ids.size)

    shape = property(lambda self: (self.n_rows, self.rows.shape[1]))
    device = property(lambda self: self.rows.device)
    dtype = property(lambda self: self.rows.dtype)
    requires_grad = property(lambda self: self.rows.requires_grad)

    def dim(self):
        return 2

    def dense(self):
        cnt = (self.run_off[1:] - self.run_off[:-1]).long()
        return self.rows.repeat_interleave(cnt, dim=0, output_size=self.n_rows)

    def to(self, device):
        rows = self.rows.to(device)
        return RepeatedRows(rows, self.run_off.to(rows.device), self.n_rows)


def dense_rows(x):
    """a plain tensor from a tensor or a RepeatedRows"""
    return x.dense() if isinstance(x, RepeatedRows) else x


class BilinearRunsFunction(torch.autograd.Function):
    """s_i = e1_i^T W q_i (exp optionally) for queries given as RepeatedRows: V = rows W^T has U rows, backward's weight gradient
    sums a run's pairs first (txe_bilinear_runs_*).  No gradient to the queries."""

    @staticmethod
    def forward(ctx, e1, W, apply_exp, rows, run_off):
        _need_cuda(e1, rows, W)
        e1, ld1 = _rows(e1)
        rows, ldq = _rows(rows)
        Wf = _f32(W).reshape(W.shape[-2], W.shape[-1])
        G, l = e1.shape
        U, r = rows.shape
        s = _empty((G,), e1)
        V = _empty((max(U, 1), l), e1)
        with _lib.on_device(e1.device):
            call("txe_bilinear_runs_fwd", ptr(e1), ld1, ptr(rows), ldq, ptr(run_off), G, U, l, r, ptr(Wf), int(apply_exp), ptr(V), ptr(s),
                 _lib.stream_ptr())
        ctx.misc = (e1, ld1, rows, ldq, run_off, V, s, int(apply_exp), W.shape)
        return s.unsqueeze(1)

    @staticmethod
    def backward(ctx, ds):
        e1, ld1, rows, ldq, run_off, V, s, apply_exp, wshape = ctx.misc
        G, l = e1.shape
        U, r = rows.shape
        ds = _f32(ds.reshape(-1))
        d_e1 = _empty((G, l), e1)
        dW = _empty((l, r), e1)
        with _lib.on_device(e1.device):
            wsb = pure("txe_bilinear_runs_bwd_ws_bytes", U, l, r)
            ws = _ws(wsb, e1)
            call("txe_bilinear_runs_bwd", ptr(e1), ld1, ptr(rows), ldq, ptr(run_off), G, U, l, r, apply_exp, ptr(V), ptr(s), ptr(ds), ptr(d_e1), l,
                 ptr(dW), ptr(ws), wsb, _lib.stream_ptr())
        return d_e1, dW.reshape(wshape), None, None, None


def find_row_runs(e2):
    """runs of equal consecutive rows of the stacked query matrix e2 [G, r], found on the device (txe_rows_find_runs): returns
    (run_id [G], run_off [G + 1], n_runs [1]) int32 device tensors -- no host synchronisation"""
    _need_cuda(e2)
    e2c, ld2 = _rows(e2)
    G, r = e2c.shape
    run_id = torch.empty(max(G, 1), dtype=torch.int32, device=e2.device)
    run_off = torch.empty(G + 1, dtype=torch.int32, device=e2.device)
    n_runs = torch.empty(1, dtype=torch.int32, device=e2.device)
    with _lib.on_device(e2.device):
        call("txe_rows_find_runs", ptr(e2c), ld2, G, r, ptr(run_id), ptr(run_off), ptr(n_runs), _lib.stream_ptr())
    return run_id, run_off, n_runs


class BilinearStackedRunsFunction(torch.autograd.Function):
    """BilinearRunsFunction on the reference collate's STACKED query matrix (one row per pair): the runs of equal consecutive rows are
    found on the device in every call (bit-wise row comparison + a scan, no host synchronisation) and the products run on one row per
    run (txe_bilinear_stacked_*).  Right for any e2; the caller (model_zoo._Bilinear) takes this form when its first training batch
    showed that rows repeat.  No gradient to the queries."""

    @staticmethod
    def forward(ctx, e1, e2, W, apply_exp):
        _need_cuda(e1, e2, W)
        e1, ld1 = _rows(e1)
        e2, ld2 = _rows(e2)
        Wf = _f32(W).reshape(W.shape[-2], W.shape[-1])
        G, l = e1.shape
        r = e2.shape[1]
        _run_id, run_off, n_runs = find_row_runs(e2)
        s = _empty((G,), e1)
        V = _empty((max(G, 1), l), e1)                        # (sized for G runs; the batch's runs fill the first rows)
        with _lib.on_device(e1.device):
            call("txe_bilinear_stacked_fwd", ptr(e1), ld1, ptr(e2), ld2, ptr(run_off), ptr(n_runs), G, l, r, ptr(Wf), int(apply_exp), ptr(V),
                 ptr(s), _lib.stream_ptr())
        ctx.misc = (e1, ld1, e2, ld2, run_off, n_runs, V, s, int(apply_exp), W.shape)
        return s.unsqueeze(1)

    @staticmethod
    def backward(ctx, ds):
        e1, ld1, e2, ld2, run_off, n_runs, V, s, apply_exp, wshape = ctx.misc
        G, l = e1.shape
        r = e2.shape[1]
        ds = _f32(ds.reshape(-1))
        d_e1 = _empty((G, l), e1)
        dW = _empty((l, r), e1)
        with _lib.on_device(e1.device):
            wsb = pure("txe_bilinear_stacked_bwd_ws_bytes", G, l, r)
            ws = _ws(wsb, e1)
            call("txe_bilinear_stacked_bwd", ptr(e1), ld1, ptr(e2), ld2, ptr(run_off), ptr(n_runs), G, l, r, apply_exp, ptr(V), ptr(s), ptr(ds),
                 ptr(d_e1), l, ptr(dW), ptr(ws), wsb, _lib.stream_ptr())
        return d_e1, None, dW.reshape(wshape), None


def gcn_folded_graph_vector_ok(csr, cfg, params):
    """may a GCN stack hand out Z instead of hg (cfg.final = 'collapse_z')?  The output layer's bias then rides as one more weight row
    behind a column of Z that counts as 1: its input width must leave a padding column ((Kin) % 32 != 0)."""
    if _NO_MATCH_FOLD or csr.n_edges <= 0 or csr.n_graphs <= 0 or csr.n_nodes <= 0:
        return False
    W, b = params[-3], params[-2]
    return b is None or (W.shape[0] % 32) != 0


def folded_graph_vector_ok(csr, cfg):
    """may a GAT stack hand out Z instead of hg (cfg.final = 'collapse_z')?  Needs the fused backward of the folded layer (the only one
    that takes dZ): at least two layers, the shapes txe_gat_fused_bwd_supported covers, edges, and the default routes."""
    if _NO_MATCH_FOLD or _NO_FUSED_BWD or cfg.n_layers < 2 or cfg.heads[-1] != 1 or csr.n_edges <= 0 or csr.n_graphs <= 0 or csr.n_nodes <= 0:
        return False
    kh = cfg.heads[-2] * cfg.out_dims[-2]
    return pure("txe_gat_fused_bwd_supported", kh, cfg.pos_dims[-1], cfg.heads[-2], cfg.out_dims[-2]) == 1


def folded_graph_linear(Z, Wp, D, link=None):
    """hg [G, D] = Z [G, Kp] Wp[:D]^T (link.by_k: Z Wp[:, :D] + bias), outside autograd (DeferredGraphVector.detach)"""
    _need_cuda(Z, Wp)
    G, Kp = Z.shape
    hg = _empty((G, D), Z)
    with _lib.on_device(Z.device):
        tws = _tail_ws(Z)
        if link is not None and link.by_k:          # a GCN layer's packing: hg = Z Wp[:Kp, :D] + bias (row one_col; Z's column there is 0)
            call("txe_gemm_plain", 1, ptr(Z), Kp, ptr(Wp), Wp.stride(0), ptr(hg), D, G, D, Kp, 1, 0, ptr(tws), tws.numel(), _lib.stream_ptr())
            if link.one_col >= 0:
                hg += Wp[link.one_col, :D]
        else:
            call("txe_gemm_plain", 0, ptr(Z), Kp, ptr(Wp), Kp, ptr(hg), D, G, D, Kp, 1, 0, ptr(tws), tws.numel(), _lib.stream_ptr())
    return hg


class FoldedGraphLinearFunction(torch.autograd.Function):
    """hg [G, D] = Z [G, Kp] Wp[:D]^T -- the graph vector of a 'collapse_z' stack materialised after all (some consumer other than the
    bilinear run matcher wants the tensor).  Backward: dZ = d_hg Wp[:D] through autograd, the weight gradient's main part d_hg^T Z as
    split-K slices through the FoldLink (the stack's backward finishes dW)."""

    @staticmethod
    def forward(ctx, Z, Wp, link, D):
        _need_cuda(Z, Wp)
        G, Kp = Z.shape
        hg = folded_graph_linear(Z, Wp, D, link)
        note_route("fold", "materialised")
        ctx.misc = (Z, Wp, link, D)
        return hg

    @staticmethod
    def backward(ctx, d_hg):
        Z, Wp, link, D = ctx.misc
        G, Kp = Z.shape
        d_hg, ld = _rows(_f32(d_hg))
        dZ = _empty((G, Kp), Z)
        if link.by_k:                               # dZ = d_hg Wp[:, :D]^T;  part [Kp][D] = Z^T d_hg, row one_col = d_bias = column sums of d_hg
            part = _empty((Kp, D), Z)
            with _lib.on_device(Z.device):
                tws = _tail_ws(Z)
                call("txe_gemm_plain", 0, ptr(d_hg), ld, ptr(Wp), Wp.stride(0), ptr(dZ), Kp, G, Kp, D, 1, 0, ptr(tws), tws.numel(), _lib.stream_ptr())
                call("txe_gemm_plain", 2, ptr(Z), Kp, ptr(d_hg), ld, ptr(part), D, Kp, D, G, 1, 0, None, 0, _lib.stream_ptr())
            if link.one_col >= 0:
                part[link.one_col] = d_hg.sum(0)
            link.part, link.S = part, 1
            return dZ, None, None, None
        S = max(1, min(8, G // 512))
        part = _empty((S * D, Kp), Z)
        with _lib.on_device(Z.device):
            tws = _tail_ws(Z)
            call("txe_gemm_plain", 1, ptr(d_hg), ld, ptr(Wp), Kp, ptr(dZ), Kp, G, Kp, D, 1, 0, ptr(tws), tws.numel(), _lib.stream_ptr())
            call("txe_gemm_plain", 2, ptr(d_hg), ld, ptr(Z), Kp, ptr(part), Kp, D, Kp, G, S, 0, None, 0, _lib.stream_ptr())
        link.part, link.S = part, S
        return dZ, None, None, None


def folded_match_job(e2, rows, run_off, Wm):
    """What a 'collapse_z' stack calls right before its Z sweep (cfg.fold_job): the query-side half of BilinearFoldedRunsFunction -- the
    runs (found on the device in the stacked e2, or given as rows + run_off), V = Wm q, T = Wp[:D]^T V and the graph -> run map -- on the
    caller's stream.  T then rides in the sweep (FoldLink.e_part) and the matcher's forward starts from the scores."""
    def job(Wp, D):
        Wmf = _f32(Wm).reshape(Wm.shape[-2], Wm.shape[-1])
        l, r = Wmf.shape
        if l != D or not Wm.is_cuda:
            return None
        Kp = Wp.shape[1]
        with _lib.on_device(Wp.device):
            if rows is None:
                Q, ldq = _rows(e2)
                G = U = Q.shape[0]
                run_id, roff, n_runs = find_row_runs(Q)
                first_row = 1
            else:
                Q, ldq = _rows(rows)
                U, first_row, roff, n_runs = Q.shape[0], 0, run_off, None
                G = None
            V, T = _empty((max(U, 1), l), Wp), _empty((max(U, 1), Kp), Wp)
            call("txe_bilinear_folded_fwd", None, Kp, G if G is not None else 1, Kp, ptr(Wp), Kp, l, ptr(Q), ldq, r, ptr(roff), ptr(n_runs), U, first_row,
                 ptr(Wmf), 0, ptr(V), ptr(T), None, 1, 0, -1, _lib.stream_ptr())
        fw = dict(e2=e2, rows=rows, run_off_in=run_off, Wm=Wm, Wm_version=Wm._version, Wp=Wp, Q=Q, ldq=ldq, roff=roff, n_runs=n_runs, U=U,
                  first_row=first_row, V=V, T=T, run_id=(run_id if rows is None else None))
        return fw
    return job


def _fold_job_run_ids(fw, G, ref):
    """graph -> run for the given-runs form (the stacked form's run detection has produced it)"""
    if fw["run_id"] is None:
        rid = torch.empty(max(G, 1), dtype=torch.int32, device=ref.device)
        call("txe_runs_expand", ptr(fw["roff"]), fw["U"], G, ptr(rid), _lib.stream_ptr())
        fw["run_id"] = rid
    return fw["run_id"]


class BilinearFoldedRunsFunction(torch.autograd.Function):
    """The bilinear match on the folded graph vector (txe_bilinear_folded_*): s_i = <Z_i, T[u(i)]>, T[u] = Wp[:D]^T (Wm q_u) -- the output
    layer's D x Kp product runs on the U run rows of the repeating queries instead of the G graph rows, forward and backward.  Queries:
    the stacked matrix e2 [G, r] (runs found on the device, rows = run_off = None) or the U distinct rows + run offsets.  Gradients: dZ
    (to the stack through autograd), the main part of the output layer's dW (through the FoldLink), dWm.  None to the queries."""

    @staticmethod
    def forward(ctx, Z, Wp, link, D, Wm, apply_exp, e2, rows, run_off):
        _need_cuda(Z, Wp, Wm)
        G, Kp = Z.shape
        Wmf = _f32(Wm).reshape(Wm.shape[-2], Wm.shape[-1])
        l, r = Wmf.shape
        if l != D:
            raise RuntimeError("bilinear matcher: l_dim does not match the graph vector")
        fw = link.fwd
        ready = (fw is not None and fw["Wp"] is Wp and fw["Wm"] is Wm and fw["Wm_version"] == Wm._version and fw["e2"] is e2
                 and fw["rows"] is rows and fw["run_off_in"] is run_off and (rows is not None or fw["U"] == G))
        if ready:                                   # the stack asked for the runs, V and T before its Z sweep (folded_match_job)
            Q, ldq, run_off, n_runs, U, first_row, V, T = (fw[k] for k in ("Q", "ldq", "roff", "n_runs", "U", "first_row", "V", "T"))
        else:
            link.fwd = link.e_part = None           # (whatever rode in the sweep belongs to other queries / weights)
            if rows is None:
                Q, ldq = _rows(e2)
                _run_id, run_off, n_runs = find_row_runs(Q)
                U, first_row = G, 1
            else:
                Q, ldq = _rows(rows)
                n_runs, U, first_row = None, Q.shape[0], 0
            V, T = _empty((max(U, 1), l), Z), _empty((max(U, 1), Kp), Z)
        s = _empty((G,), Z)
        # 'edot': the stack ran the matcher's job and its Z sweep carried T; 'job': it ran the job only; 'inline': V / T formed here
        note_route("fold", "edot" if (ready and link.e_part is not None and "score" in fw) else ("job" if ready else "inline"))
        with _lib.on_device(Z.device):
            if ready and link.e_part is not None and "score" in fw:
                # T rode in the stack's Z sweep: the scores are sums of its per-node dot products over each graph's few nodes, no sweep over Z
                goff, n_, g_, kh_, pd_, coef_, wsum_, fp_, masked_ = fw["score"]
                call("txe_gat_collapse_fold_scores", ptr(goff), n_, g_, kh_, pd_, ptr(coef_), ptr(wsum_), ptr(link.e_part), fp_, masked_, int(apply_exp),
                     ptr(s), _lib.stream_ptr())
            else:
                call("txe_bilinear_folded_fwd", ptr(Z), Kp, G, Kp, ptr(Wp), Wp.stride(0), l, ptr(Q), ldq, r, ptr(run_off), ptr(n_runs), U, first_row,
                     ptr(Wmf), int(apply_exp), ptr(V), ptr(T), ptr(s), 2 if ready else 3, int(link.by_k), int(link.one_col), _lib.stream_ptr())
        ctx.misc = (Z, Wp, link, Wmf, Q, ldq, run_off, n_runs, U, first_row, V, T, s, int(apply_exp), Wm.shape)
        return s.unsqueeze(1)

    @staticmethod
    def backward(ctx, ds):
        Z, Wp, link, Wmf, Q, ldq, run_off, n_runs, U, first_row, V, T, s, apply_exp, wshape = ctx.misc
        G, Kp = Z.shape
        l, r = Wmf.shape
        ds = _f32(ds.reshape(-1))
        edot = link.e_part is not None              # the stack reads "dZ[g]" as dsl_g T[run(g)] (FoldLink): no dZ tensor exists
        dZ = None if edot else _empty((G, Kp), Z)
        dT, dV = _empty((max(U, 1), Kp), Z), _empty((max(U, 1), l), Z)
        dWm, dWf = _empty((l, r), Z), (_empty((Kp, l), Z) if link.by_k else _empty((l, Kp), Z))
        with _lib.on_device(Z.device):
            call("txe_bilinear_folded_bwd", ptr(Z), Kp, G, Kp, ptr(Wp), Wp.stride(0), l, ptr(Q), ldq, r, ptr(run_off), ptr(n_runs), U, first_row, apply_exp,
                 ptr(V), ptr(T), ptr(s), ptr(ds), ptr(dZ), Kp, ptr(dT), ptr(dV), ptr(dWm), ptr(dWf), int(link.by_k), int(link.one_col),
                 _lib.stream_ptr())
        link.part, link.S = dWf, 1
        link.m = (ds, s, apply_exp) if link.e_part is not None else None
        return dZ, None, None, None, dWm.reshape(wshape), None, None, None, None


# ================================================================================================================
# inference-side helpers (no autograd)
# ================================================================================================================
def bilinear_project(hg, W):
    """U = hg @ W[0]  (G x r): the factored half of the bilinear form, computed once per candidate set."""
    _need_cuda(hg, W)
    hg, ld = _rows(hg)
    Wf = _f32(W).reshape(W.shape[-2], W.shape[-1])
    G, l = hg.shape
    r = Wf.shape[1]
    # rows zero-padded to a whole number of 32-column k-tiles: the scoring GEMM then runs every k-tile on the plain 16-byte
    # loader (r = 250 would leave a ragged last tile on the generic one); zeros add exactly nothing to the products
    rp = (r + 31) // 32 * 32
    Ufull = _empty((max(G, 1), rp), hg)
    U = Ufull[:G, :r]
    if rp != r:
        # the WEIGHT is padded instead of U: W [l][r] -> [l][rp] with zero columns (0.5 MB), so the product writes U's zero padding
        # itself and reads a 16-byte-aligned operand (rows of 250 floats are only 8-byte aligned: the two-float loader ran this GEMM
        # at 47 TF/s, 130 us of the 0.54-ms MAG-CS scoring pass)
        Wp = torch.zeros((l, rp), dtype=torch.float32, device=hg.device)
        Wp[:, :r].copy_(Wf)
        Wf = Wp
    with _lib.on_device(hg.device):
        swb = 0 if (_NO_SPLIT_GEMM or G < 1) else pure("txe_gemm_plain_split_ws_bytes", G, rp, l)
        sws = _ws(swb, hg) if swb else None
        call("txe_bilinear_project", ptr(hg), ld, G, l, ptr(Wf), rp, ptr(Ufull), rp, ptr(sws), swb, _lib.stream_ptr())
    _ZERO_PADDED[U.data_ptr()] = (rp, weakref.ref(Ufull))
    if not _NO_SPLIT_GEMM and G >= 1:
        # the candidates' bf16 planes for the scoring loop, once per candidate set (every score_* call on this U finds them: the loop
        # over query blocks packs only its queries).  U is never written after this point -- the planes ARE this U.
        with _lib.on_device(hg.device):
            planes = torch.empty(pure("txe_split_packed_bytes", G, rp), dtype=torch.uint8, device=hg.device)
            call("txe_split_pack", ptr(Ufull), rp, G, rp, 1, ptr(planes), _lib.stream_ptr())
        key = U.data_ptr()
        _PACKED_U[key] = (rp, G, weakref.ref(Ufull, lambda _ref, key=key: _PACKED_U.pop(key, None)), planes)    # (the planes die with their U)
    return U


_ZERO_PADDED = {}      # data_ptr of a U made by bilinear_project -> (zero-padded row width, weak reference to its storage)
_PACKED_U = {}         # ... -> (contraction width, rows, weak reference to its storage, the packed bf16 planes of side 1)


def _u_planes(U, r):
    """the planes bilinear_project packed for this very U (all its rows, contraction width r), or None"""
    ent = _PACKED_U.get(U.data_ptr())
    if ent is None or _NO_SPLIT_GEMM:
        return None
    rp, G, ref, planes = ent
    if ref() is None:
        del _PACKED_U[U.data_ptr()]
        return None
    return planes if (rp == r and U.shape[0] == G and U.stride(0) == rp) else None


def _padded_width(U, r):
    """row width up to which U's columns beyond r are known zeros (bilinear_project output), else r"""
    ent = _ZERO_PADDED.get(U.data_ptr())
    if ent is None:
        return r
    rp, ref = ent
    if ref() is None:                                   # the buffer died; the address may have been reused
        del _ZERO_PADDED[U.data_ptr()]
        return r
    return rp if (U.stride(0) == rp and U.shape[1] == r) else r


def gather_padded_rows(U, idx):
    """U[idx] for a bilinear_project output, gathered WITH its zero padding: the result is again a [n, r] view of a [n, rp] buffer that
    score kernels take on the plain 16-byte loader (a plain U[idx] is a packed [n, 250] copy: 8-byte rows, a ragged last k-tile)"""
    ent = _ZERO_PADDED.get(U.data_ptr())
    full = ent[1]() if ent is not None else None
    if full is None or U.stride(0) != ent[0]:
        return U.index_select(0, idx)
    out_full = full.index_select(0, idx)
    out = out_full[:, :U.shape[1]]
    if len(_ZERO_PADDED) > 256:                         # (entries of buffers that died)
        for k in [k for k, v in _ZERO_PADDED.items() if v[1]() is None]:
            del _ZERO_PADDED[k]
    _ZERO_PADDED[out.data_ptr()] = (ent[0], weakref.ref(out_full))
    return out


def _pad_queries(Q, r, rp):
    """queries on the same zero-padded pitch as U"""
    Qp = torch.zeros((Q.shape[0], rp), dtype=torch.float32, device=Q.device)
    Qp[:, :r].copy_(Q)
    return Qp


def _score_sws(nq, G, r, ref, u_packed=False):
    """scratch with which a scoring entry point runs on the bf16 matrix pipe: (tensor, bytes), or (None, 0) on the fp32-MFMA
    route.  The four entry points compare scores bit for bit among themselves: the switch is one module attribute for all of them.
    u_packed: the candidates' planes exist already (_u_planes) -- room for the queries' only."""
    if _NO_SPLIT_GEMM or nq < 1 or G < 1:
        return None, 0
    n = (pure("txe_split_packed_bytes", int(nq), int(r)) + 255) // 256 * 256 if u_packed else pure("txe_score_split_ws_bytes", int(nq), int(G), int(r))
    return _ws(n, ref), n


def score_block(Q, U, apply_exp, out=None):
    """S[q][g] = match(hg[g], Q[q]) for a block of queries against every candidate (test_fast.py:116-123)."""
    _need_cuda(Q, U)
    Q, ldq = _rows(Q)
    U, ldu = _rows(U)
    nq, r = Q.shape
    rp = _padded_width(U, r)
    if rp != r and nq > 0:                              # both operands zero-padded to whole k-tiles: K = rp
        Q = _pad_queries(Q, r, rp)
        ldq, r = rp, rp
    elif ldq % 4 != 0 and nq > 0:                       # query rows re-laid out on a 16-byte pitch (a few hundred KB per block)
        Qp = _empty((nq, (r + 3) // 4 * 4), Q)[:, :r]
        Qp.copy_(Q)
        Q, ldq = Qp, Qp.stride(0)
    G = U.shape[0]
    S = out if out is not None else _empty((nq, (G + 3) // 4 * 4), Q)[:, :G]     # 16-byte row pitch: vector stores / rank sweeps
    with _lib.on_device(Q.device):
        tws = _tail_ws(Q)
        up = _u_planes(U, r)
        sws, swb = _score_sws(nq, G, r, Q, up is not None)
        call("txe_score_block", ptr(Q), ldq, nq, ptr(U), ldu, G, r, int(apply_exp), ptr(S), S.stride(0), ptr(tws), tws.numel(), ptr(sws), swb,
             ptr(up), _lib.stream_ptr())
    return S


def positive_scores(Q, U, apply_exp, pos_off, pos_idx):
    """thr[j] = match(hg[pos_idx[j]], Q[q(j)]) for each query's true parents, through the SAME score kernel as the full block
    (bit-identical values: the fused ranking compares against them).  pos_idx rows of U that are out of range (< 0: a positive that
    lives in another candidate shard) give 0."""
    _need_cuda(Q, U)
    dev = Q.device
    pos_off = _i32(pos_off, dev)
    pos_idx = _i32(pos_idx, dev)
    n_pos = int(pos_idx.numel())
    if n_pos == 0:
        return torch.zeros(0, dtype=torch.float32, device=dev)
    counts = (pos_off[1:] - pos_off[:-1]).long()
    qid = torch.repeat_interleave(torch.arange(Q.shape[0], device=dev), counts)
    local = pos_idx >= 0
    Ug = U[pos_idx.clamp(min=0).long()]
    Sp = score_block(Q, Ug, apply_exp)
    thr = Sp[qid, torch.arange(n_pos, device=dev)]
    return torch.where(local, thr, torch.zeros_like(thr)).contiguous()


def pad_queries_like(Q, U):
    """the query matrix zero-padded to U's k-tile pitch, ONCE for a whole scoring loop: blocks `Qp[q0:q1, :r]` of the result go to
    score_count_block(..., q_padded=True) without being copied again.  Returns Q itself when U carries no padding."""
    r = Q.shape[1]
    rp = _padded_width(_rows(U)[0], r)
    Qr = _rows(Q)[0]                                    # fp32, unit column stride: score_count_block(q_padded=True) takes the pointer as it is
    return _pad_queries(Qr, r, rp)[:, :r] if rp != r else Qr


def positive_scores_staircase(Q, Up, apply_exp, pos_off, out):
    """out[j] = match(Q[q], Up[j]) for j in [pos_off[q], pos_off[q+1]): Up holds the candidate rows of the queries' true parents, query
    by query.  The score kernel's own tiles (bit-identical values), only those along the staircase (txe_score_positives)."""
    _need_cuda(Q, Up)
    ldq = Q.stride(0)
    Up, ldu = _rows(Up)
    assert Q.dtype == torch.float32 and Q.stride(1) == 1 and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= Up.shape[0]
    r = Q.shape[1]
    rp = _padded_width(Up, r)
    if rp != r and ldq == rp:            # both operands carry zeros up to the same k-tile pitch (pad_queries_like / gather_padded_rows): the
        r = rp                           # whole reduction on the plain loader -- and the very k-tiles txe_score_count_block runs
    with _lib.on_device(Q.device):
        sws, swb = _score_sws(Q.shape[0], Up.shape[0], r, Q)
        call("txe_score_positives", ptr(Q), ldq, Q.shape[0], ptr(Up), ldu, Up.shape[0], r, int(apply_exp), ptr(pos_off), ptr(out),
             ptr(sws), swb, _lib.stream_ptr())
    return out


def score_count_block(Q, U, apply_exp, pos_off, thr, larger_is_better=True, counts=None, q_padded=False):
    """fused scoring + ranking of one query block against a candidate (shard) matrix U: int32 counts [n_pos] of candidates that beat
    each positive's score thr[j] (txe_score_count_block; no [nq x G] score block is materialised).
    q_padded: Q is a row block of pad_queries_like(all queries, U) -- its columns up to U's padded width are zeros already."""
    _need_cuda(Q, U)
    if q_padded:
        assert Q.dtype == torch.float32 and Q.dim() == 2 and Q.stride(1) == 1, "q_padded: a row block of pad_queries_like()"
        ldq = Q.stride(0)
    else:
        Q, ldq = _rows(Q)
    U, ldu = _rows(U)
    nq, r = Q.shape
    rp = _padded_width(U, r)
    if q_padded and rp != r and nq > 0:
        assert ldq == rp and Q.stride(1) == 1, "q_padded: a row block of pad_queries_like()"
        r = rp
    elif rp != r and nq > 0:
        Q = _pad_queries(Q, r, rp)
        ldq, r = rp, rp
    elif ldq % 4 != 0 and nq > 0:
        Qp = _empty((nq, (r + 3) // 4 * 4), Q)[:, :r]
        Qp.copy_(Q)
        Q, ldq = Qp, Qp.stride(0)
    pos_off = _i32(pos_off, Q.device)
    thr = _f32(thr)
    if counts is None:
        counts = torch.zeros(max(int(thr.numel()), 1), dtype=torch.int32, device=Q.device)
    if thr.numel() == 0:                                # a query block without a single positive: nothing to count
        return counts
    assert counts.dtype == torch.int32 and counts.is_contiguous() and counts.numel() >= thr.numel()
    with _lib.on_device(Q.device):
        up = _u_planes(U, r)
        sws, swb = _score_sws(nq, U.shape[0], r, Q, up is not None)
        call("txe_score_count_block", ptr(Q), ldq, nq, ptr(U), ldu, U.shape[0], r, int(apply_exp), ptr(pos_off), ptr(thr),
             int(larger_is_better), ptr(counts), ptr(sws), swb, ptr(up), _lib.stream_ptr())
    return counts


def score_topk_block(Q, U, apply_exp, k, larger_is_better=True, idx_base=0, q_padded=False, scratch=None):
    """fused scoring + best-k selection of one query block against a candidate (shard) matrix U (txe_score_topk_block): returns
    (idx int32 [nq, k] = candidate rows + idx_base, best first, ties by ascending row like Python's stable sort, NaN last;
    key fp32 [nq, k] = the scores, negated when smaller is better).  No [nq x G] block is materialised.  1 <= k <= min(8, G).
    q_padded: as in score_count_block.  scratch: dict reused across the blocks of a loop (the per-tile lists)."""
    _need_cuda(Q, U)
    if q_padded:
        assert Q.dtype == torch.float32 and Q.dim() == 2 and Q.stride(1) == 1, "q_padded: a row block of pad_queries_like()"
        ldq = Q.stride(0)
    else:
        Q, ldq = _rows(Q)
    U, ldu = _rows(U)
    nq, r = Q.shape
    G = U.shape[0]
    assert 1 <= k <= 8 and k <= G, "score_topk_block: 1 <= k <= min(8, candidates)"
    rp = _padded_width(U, r)
    if q_padded and rp != r and nq > 0:
        assert ldq == rp
        r = rp
    elif rp != r and nq > 0:
        Q = _pad_queries(Q, r, rp)
        ldq, r = rp, rp
    elif ldq % 4 != 0 and nq > 0:
        Qp = _empty((nq, (r + 3) // 4 * 4), Q)[:, :r]
        Qp.copy_(Q)
        Q, ldq = Qp, Qp.stride(0)
    idx = torch.empty((nq, k), dtype=torch.int32, device=Q.device)
    key = _empty((nq, k), Q)
    if nq == 0:
        return idx, key
    with _lib.on_device(Q.device):
        nt = pure("txe_score_topk_tiles", G)
        up = _u_planes(U, r)
        sws, swb = _score_sws(nq, G, r, Q, up is not None)
        need = nq * nt * k
        sc = scratch if scratch is not None else {}
        if sc.get("n", 0) < need or sc.get("nq", 0) < nq or sc["key"].device != Q.device:
            sc["key"], sc["idx"], sc["n"] = _empty((need,), Q), torch.empty(need, dtype=torch.int32, device=Q.device), need
            sc["floor"], sc["nq"] = torch.empty(nq, dtype=torch.int32, device=Q.device), nq
        call("txe_score_topk_block", ptr(Q), ldq, nq, ptr(U), ldu, G, r, int(apply_exp), int(larger_is_better), int(k), int(idx_base),
             ptr(sc["key"]), ptr(sc["idx"]), ptr(sc["floor"]), ptr(idx), ptr(key), ptr(sws), swb, ptr(up), _lib.stream_ptr())
    return idx, key


def topk_merge(keys, idx, k):
    """best k of the (key, idx) entries of every row (txe_topk_merge; keys fp32 / idx int32 [nq, cnt], idx == INT_MAX: empty slot):
    the merge of per-rank best-k lists of a candidate-sharded loop.  Returns (idx [nq, k], key [nq, k])."""
    _need_cuda(keys, idx)
    keys, idx = _f32(keys), idx.to(torch.int32).contiguous()
    nq, cnt = keys.shape
    out_i = torch.empty((nq, k), dtype=torch.int32, device=keys.device)
    out_k = _empty((nq, k), keys)
    with _lib.on_device(keys.device):
        call("txe_topk_merge", ptr(keys), ptr(idx), nq, cnt, int(k), 0, ptr(out_i), ptr(out_k), _lib.stream_ptr())
    return out_i, out_k


def rank_finalize(pos_off, thr, counts, larger_is_better=True, out=None):
    """ranks (int32) from the fused counts: positives never count against each other (metric.py:7-31).  out: int32 [>= n_pos]"""
    _need_cuda(thr)
    pos_off = _i32(pos_off, thr.device)
    n_pos = int(thr.numel())
    ranks = out if out is not None else torch.empty(max(n_pos, 1), dtype=torch.int32, device=thr.device)
    if n_pos == 0:
        return ranks[:0]
    with _lib.on_device(thr.device):
        call("txe_rank_finalize", ptr(pos_off), int(pos_off.numel()) - 1, ptr(_f32(thr)), ptr(counts), int(larger_is_better), ptr(ranks),
             _lib.stream_ptr())
    return ranks[:n_pos]


def rank_block(S, pos_off, pos_idx, larger_is_better=True):
    """ranks of each query's true parents among the candidates (metric.py:7-31 semantics), int32 on device."""
    _need_cuda(S)
    nq, G = S.shape
    pos_off = _i32(pos_off, S.device)
    pos_idx = _i32(pos_idx, S.device)
    ranks = torch.empty(max(int(pos_idx.numel()), 1), dtype=torch.int32, device=S.device)
    with _lib.on_device(S.device):
        call("txe_rank_block", ptr(S), S.stride(0), nq, G, ptr(pos_off), ptr(pos_idx), ptr(ranks), int(larger_is_better), None,
             _lib.stream_ptr())
    return ranks[:pos_idx.numel()]
