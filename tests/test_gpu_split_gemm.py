"""fp32 products on the bf16 matrix pipe (csrc/txe_gemm_split.*): the packed three-plane operands are EXACT
(x1 + x2 + x3 == x bit for bit), the NT and TN products agree with float64 at least as well as an fp32 product does, on ragged shapes,
and the model's first-layer projection / weight gradient take the route."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


# max |err| / sum_k |a||b| an fp32 product may show: fp32 kernels themselves span 0.7e-7 .. 6.4e-7 on these shapes (torch.mm picks
# different kernels / summation orders by shape); the bf16-pipe products measure 0.4e-7 .. 4.5e-7
ABS_FLOOR = 6e-7


def _bf16_bits_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def _slot_row(side, rb, s):
    """csrc/txe_gemm_split.h split_slot_row"""
    return 32 * rb + s if side == 0 else 128 * (rb >> 2) + 64 * ((rb & 3) >> 1) + 2 * s + (rb & 1)


@pytest.mark.parametrize("rows,cols,side", [(300, 50, 0), (129, 33, 1), (1, 1, 0), (257, 320, 1)])
def test_packed_planes_sum_to_the_operand_exactly(rows, cols, side):
    from taxoexpan_amd import _lib
    g = torch.Generator().manual_seed(rows * 7 + cols)
    x = torch.randn(rows, cols, generator=g) * torch.exp(4 * torch.randn(rows, cols, generator=g))
    x[0, 0] = 0.0
    xd = x.to(_dev())
    nb = _lib.call("txe_split_packed_bytes", rows, cols)
    buf = torch.zeros(nb, dtype=torch.uint8, device=_dev())
    _lib.call("txe_split_pack", xd.data_ptr(), cols, rows, cols, side, buf.data_ptr(), _lib.stream_ptr())
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().view(np.uint16)
    nkt = (cols + 15) // 16
    nrb = ((rows + 767) // 768) * 24
    frag = raw.reshape(nrb, nkt, 3, 2, 32, 8)          # [rb][kt][plane][kh][slot][8 k]
    total = np.zeros((rows, nkt * 16))
    seen = np.zeros((rows, nkt * 16), dtype=bool)
    for rb in range(nrb):
        for s in range(32):
            r = _slot_row(side, rb, s)
            if r >= rows:
                assert not frag[rb, :, :, :, s, :].any()          # padding rows are zeros
                continue
            for kt in range(nkt):
                for kh in range(2):
                    k0 = kt * 16 + kh * 8
                    total[r, k0:k0 + 8] = sum(_bf16_bits_to_f64(frag[rb, kt, p, kh, s, :]) for p in range(3))
                    seen[r, k0:k0 + 8] = True
    assert seen.all()
    np.testing.assert_array_equal(total[:, :cols], x.numpy().astype(np.float64))     # exact: three bf16 numbers carry the 24 bits
    assert not total[:, cols:].any()


RAW_MARK = 0x7FC0         # csrc/txe_gemm_split.h SPL_RAW_MARK: plane 3 of a fragment stored raw (it holds an exceptional element)


def _decode_fragment(frag_rb_kt):
    """[plane][kh][slot][8] uint16 -> float64 values [kh][slot][8] and whether the fragment is stored raw"""
    p1, p2, p3 = frag_rb_kt[0], frag_rb_kt[1], frag_rb_kt[2]
    if (p3 == RAW_MARK).all():
        bits = (p1.astype(np.uint32) << 16) | p2.astype(np.uint32)
        return bits.view(np.float32).astype(np.float64), True
    assert not (p3 == RAW_MARK).any()
    return _bf16_bits_to_f64(p1) + _bf16_bits_to_f64(p2) + _bf16_bits_to_f64(p3), False


def _exceptional(x):
    return ~np.isfinite(x) | (np.abs(x.astype(np.float64)) >= 2.0 ** 120)


@pytest.mark.parametrize("side", [0, 1])
def test_packed_form_carries_the_whole_fp32_domain(side):
    """+-Inf, NaN, 3.4e38, 2^120 and values inside the ordinary range in one operand: a fragment (32 slots x 16 columns) that holds an
    exceptional element (not finite, or |x| >= 2^120) is stored RAW (high halves, low halves, NaN plane) and decodes to the operand bit
    for bit; every other fragment is three planes whose sum is the operand exactly down to 2^-100, and within 2^-126 below that"""
    from taxoexpan_amd import _lib
    rows, cols = 200, 70
    g = torch.Generator().manual_seed(11 + side)
    x = torch.randn(rows, cols, generator=g)
    specials = [float("inf"), float("-inf"), float("nan"), 1e-45, -3e-39, 1.1754944e-38, 1e-38, 2.0 ** -101, 3.4028234e38, -3.39e38, 2.0 ** 120, 1e37, 2.0 ** -100]
    for i, v in enumerate(specials):
        x[(7 * i) % rows, (5 * i) % cols] = v
    x[150:160, 40:50] = torch.randn(10, 10, generator=g) * 1e-41            # a block of subnormals
    xd = x.to(_dev())
    buf = torch.zeros(_lib.call("txe_split_packed_bytes", rows, cols), dtype=torch.uint8, device=_dev())
    _lib.call("txe_split_pack", xd.data_ptr(), cols, rows, cols, side, buf.data_ptr(), _lib.stream_ptr())
    torch.cuda.synchronize()
    nkt, nrb = (cols + 15) // 16, ((rows + 767) // 768) * 24
    frag = buf.cpu().numpy().view(np.uint16).reshape(nrb, nkt, 3, 2, 32, 8)
    xn = x.numpy()
    pad = np.zeros((nrb * 32 + 768, nkt * 16), dtype=np.float32)
    pad[:rows, :cols] = xn
    n_raw = 0
    for rb in range(nrb):
        rws = np.array([_slot_row(side, rb, s) for s in range(32)])
        for kt in range(nkt):
            vals, raw = _decode_fragment(frag[rb, kt])
            want = np.stack([pad[rws][:, kt * 16 + 8 * kh:kt * 16 + 8 * kh + 8] for kh in range(2)])     # [kh][slot][8]
            assert raw == bool(_exceptional(want).any()), (rb, kt)
            n_raw += raw
            if raw:
                np.testing.assert_array_equal(vals.astype(np.float32).view(np.uint32), want.view(np.uint32))
            else:
                w64 = want.astype(np.float64)
                big = np.abs(w64) >= 2.0 ** -100
                np.testing.assert_array_equal(vals[big | (w64 == 0)], w64[big | (w64 == 0)])
                assert (np.abs(vals - w64) <= 2.0 ** -126).all()
    assert n_raw >= 3


def _ieee_compare(got, A, B, ref32, e_bound):
    """got (device, fp32) against the fp32 product ref32 = torch.mm on the CPU (IEEE arithmetic): the same entries are NaN, the same are
    +Inf / -Inf, and the finite ones are as close to float64 as an fp32 product (the bound of the ordinary tests, plus a few FLT_MIN:
    results in the subnormal range may be flushed by the matrix pipe as by any GPU GEMM)"""
    got, ref32 = got.cpu(), ref32.cpu()
    A64, B64 = torch.nan_to_num(A.double().cpu(), nan=0.0, posinf=0.0, neginf=0.0), torch.nan_to_num(B.double().cpu(), nan=0.0, posinf=0.0, neginf=0.0)
    # an operand element below 2^-133 may act as zero (flush-to-zero semantics below 2^-100: csrc/txe_gemm_split.h): where one meets an
    # infinity of the other operand IEEE gives +-Inf, a flushed zero gives NaN -- there only "not finite" is asked
    tiniest = lambda X64: ((X64 != 0) & (X64.abs() < 2.0 ** -132)).double()
    isinf64 = lambda X: torch.isinf(X.cpu()).double()
    loose = (tiniest(A64) @ isinf64(B) + isinf64(A) @ tiniest(B64)) > 0
    strict = ~loose
    assert not torch.isfinite(got[loose]).any()
    assert torch.equal(torch.isnan(got)[strict], torch.isnan(ref32)[strict]), (torch.isnan(got).sum().item(), torch.isnan(ref32).sum().item())
    inf = torch.isinf(ref32) & strict
    assert torch.equal(torch.isinf(got)[strict], torch.isinf(ref32)[strict]) and torch.equal(got[inf], ref32[inf])
    fin = torch.isfinite(ref32)
    ref = A64 @ B64
    scale = A64.abs() @ B64.abs()
    err = (got.double() - ref).abs()[fin]
    e32 = (ref32.double() - ref).abs()[fin]
    sc = scale[fin]
    rel32 = (e32 / sc.clamp_min(1e-300)).max().item()
    # operand elements below 2^-100 are carried to 2^-126 absolute (csrc/txe_gemm_split.h): their share of the bound
    tinyA, tinyB = ((A64 != 0) & (A64.abs() < 2.0 ** -100)).double(), ((B64 != 0) & (B64.abs() < 2.0 ** -100)).double()
    tiny = 2.0 ** -126 * (tinyA @ B64.abs() + A64.abs() @ tinyB)[fin]
    tol = max(1.5 * rel32, e_bound) * sc + tiny + 4 * 1.1754944e-38
    assert (err <= tol).all(), ((err / sc.clamp_min(1e-300)).max().item(), rel32)


def _special_operands(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) * 0.1
    A[3, 5] = float("inf")                      # Inf x finite = +-Inf along the row ...
    B[4, 5] = 0.0                               # ... Inf x 0 = NaN in this column
    A[10, :] = float("nan")
    A[20, 7] = float("-inf"); A[20, 9] = float("inf")      # Inf - Inf = NaN wherever both columns of B are nonzero of one sign ...
    A[40, :] = torch.randn(K, generator=g) * 1e-40          # a subnormal row ...
    B[6, :] = torch.randn(K, generator=g) * 1e10            # ... against a large column: products ~1e-30, every bit of the subnormals counts
    A[50, :] = torch.randn(K, generator=g).abs() * 1e-38   # just under / over FLT_MIN
    A[60, 0] = 3.0e38                                       # beyond bf16's largest number
    B[8, 0], B[9, 0], B[11, 0] = 0.5, 2.0, -2.0             # 1.5e38 (finite), +Inf, -Inf
    # ordinary-range operands whose product overflows: every term positive, +Inf in any summation order (and no OTHER pairing of the
    # scaled rows / columns overflows: whether Inf - Inf appears inside a sum depends on the order and on FMA contraction, not on IEEE)
    A[70, :] = A[70, :].abs() * 1e25; B[12, :] = B[12, :].abs() * 1e15
    B[13, 2] = float("inf"); B[14, :] = float("nan"); B[15, :] = torch.randn(K, generator=g) * 1e-42
    A[M - 1, K - 1] = float("inf")                          # the last element of a ragged tile
    return A, B


@pytest.mark.parametrize("M,N,K", [(300, 200, 100), (129, 17, 33), (1000, 260, 320)])
def test_nt_product_on_the_whole_fp32_domain(M, N, K):
    """Inf, NaN, near-FLT_MAX and overflowing operands: the bf16-pipe product gives what an IEEE fp32 product gives -- the same entries NaN,
    the same +-Inf, the finite ones inside the fp32 band; subnormal / tiny operand elements are carried to 2^-126 absolute"""
    from taxoexpan_amd import _lib
    A, B = _special_operands(M, N, K, M + K)
    Ad, Bd = A.to(_dev()), B.to(_dev())
    s = _lib.stream_ptr()
    Ap = torch.empty(_lib.call("txe_split_packed_bytes", M, K), dtype=torch.uint8, device=_dev())
    Bp = torch.empty(_lib.call("txe_split_packed_bytes", N, K), dtype=torch.uint8, device=_dev())
    C = torch.full((M, N), 123.0, device=_dev())
    _lib.call("txe_split_pack", Ad.data_ptr(), K, M, K, 0, Ap.data_ptr(), s)
    _lib.call("txe_split_pack", Bd.data_ptr(), K, N, K, 1, Bp.data_ptr(), s)
    _lib.call("txe_gemm_nt_split", Ap.data_ptr(), Bp.data_ptr(), M, N, K, C.data_ptr(), N, s)
    torch.cuda.synchronize()
    _ieee_compare(C, A, B.t(), A @ B.t(), ABS_FLOOR)


@pytest.mark.parametrize("n,M,N,S", [(500, 128, 160, 2), (333, 256, 36, 3)])
def test_tn_product_on_the_whole_fp32_domain(n, M, N, S):
    """the weight-gradient form: exceptional elements in the fp32 operand (split in the product's loader) and in the packed one"""
    from taxoexpan_amd import _lib
    At, Bt_ = _special_operands(M, N, n, n + M)          # [M][n], [N][n]: contraction over n
    A, B = At.t().contiguous(), Bt_.t().contiguous()     # A [n][M], B [n][N]
    Ad, Bd = A.to(_dev()), B.to(_dev())
    s = _lib.stream_ptr()
    Bp = torch.empty(_lib.call("txe_split_packed_t_bytes", n, N), dtype=torch.uint8, device=_dev())
    ks = (((n + S - 1) // S) + 15) // 16 * 16
    part = torch.full((S, M, N), 123.0, device=_dev())
    _lib.call("txe_split_pack_t", Bd.data_ptr(), N, n, N, Bp.data_ptr(), s)
    _lib.call("txe_gemm_tn_split", Ad.data_ptr(), M, M, Bp.data_ptr(), N, n, S, ks, part.data_ptr(), N, M * N, s)
    torch.cuda.synchronize()
    for z in range(S):
        lo, hi = min(n, z * ks), min(n, (z + 1) * ks)
        _ieee_compare(part[z], A[lo:hi].t(), B[lo:hi], A[lo:hi].t() @ B[lo:hi], ABS_FLOOR)


def test_scoring_entry_points_propagate_infinities():
    """test_fast.py:121-123 with an overflowing graph vector and an overflowing query (LBM scores reach inf legitimately:
    tests/golden/newterms.npz holds 124): the bf16-pipe scoring product gives the fp32 route's Inf / NaN / 0 pattern"""
    from taxoexpan_amd import ops
    g = torch.Generator().manual_seed(5)
    G, nq, r = 700, 130, 500
    U = (torch.randn(G, r, generator=g) * 0.05)
    Q = torch.randn(nq, r, generator=g)
    U[5, 3] = float("inf"); U[6, :] = float("nan"); U[7, 4] = float("-inf"); Q[2, 8] = float("inf"); Q[9, 3] = 0.0
    ref = Q @ U.t()
    for apply_exp in (False, True):
        S = ops.score_block(Q.to(_dev()), U.to(_dev()), apply_exp)
        torch.cuda.synchronize()
        want = ref.exp() if apply_exp else ref
        got = S.cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        assert torch.equal(torch.isinf(got), torch.isinf(want)) and torch.equal(got[torch.isinf(want)], want[torch.isinf(want)])
        fin = torch.isfinite(want)
        np.testing.assert_allclose(got[fin].numpy(), want[fin].numpy(), rtol=2e-5, atol=1e-5)


def _err(c, ref, scale):
    return ((c.double() - ref).abs() / scale).max().item()


@pytest.mark.parametrize("M,N,K", [(1000, 2008, 300), (17, 5, 3), (128, 128, 16), (391, 130, 100), (2500, 250, 512)])
def test_nt_product_against_float64(M, N, K):
    from taxoexpan_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(_dev())
    B = (torch.randn(N, K, generator=g) * 0.1).to(_dev())
    A[0, :] *= 1e15
    s = _lib.stream_ptr()
    Ap = torch.empty(_lib.call("txe_split_packed_bytes", M, K), dtype=torch.uint8, device=_dev())
    Bp = torch.empty(_lib.call("txe_split_packed_bytes", N, K), dtype=torch.uint8, device=_dev())
    ldc = N + 3
    C = torch.full((M, ldc), float("nan"), device=_dev())
    _lib.call("txe_split_pack", A.data_ptr(), K, M, K, 0, Ap.data_ptr(), s)
    _lib.call("txe_split_pack", B.data_ptr(), K, N, K, 1, Bp.data_ptr(), s)
    _lib.call("txe_gemm_nt_split", Ap.data_ptr(), Bp.data_ptr(), M, N, K, C.data_ptr(), ldc, s)
    torch.cuda.synchronize()
    assert torch.isnan(C[:, N:]).all()                       # nothing stored past N
    ref = A.double() @ B.double().t()
    scale = (A.double().abs() @ B.double().abs().t()).clamp_min(1e-300)
    e_split, e_f32 = _err(C[:, :N], ref, scale), _err(A @ B.t(), ref, scale)
    assert e_split <= max(1.5 * e_f32, ABS_FLOOR), (e_split, e_f32)      # as close to the exact product as an fp32 one


@pytest.mark.parametrize("n,M,N,S", [(1000, 128, 160, 3), (4097, 256, 320, 16), (15, 128, 160, 2), (2000, 256, 352, 4), (333, 128, 36, 2)])
def test_tn_product_against_float64(n, M, N, S):
    from taxoexpan_amd import _lib
    g = torch.Generator().manual_seed(n + M)
    A = (torch.randn(n, M, generator=g) * 0.01).to(_dev())
    B = torch.randn(n, N, generator=g).to(_dev())
    s = _lib.stream_ptr()
    Bt = torch.empty(_lib.call("txe_split_packed_t_bytes", n, N), dtype=torch.uint8, device=_dev())
    ks = (((n + S - 1) // S) + 15) // 16 * 16
    part = torch.full((S, M, N), float("nan"), device=_dev())
    _lib.call("txe_split_pack_t", B.data_ptr(), N, n, N, Bt.data_ptr(), s)
    _lib.call("txe_gemm_tn_split", A.data_ptr(), M, M, Bt.data_ptr(), N, n, S, ks, part.data_ptr(), N, M * N, s)
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()                        # slices past the last row hold zeros
    for z in range(S):
        lo, hi = min(n, z * ks), min(n, (z + 1) * ks)
        ref = A[lo:hi].double().t() @ B[lo:hi].double()
        scale = (A[lo:hi].double().abs().t() @ B[lo:hi].double().abs()).clamp_min(1e-30)
        e_split = _err(part[z], ref, scale)
        e_f32 = _err(A[lo:hi].t() @ B[lo:hi], ref, scale) if hi > lo else 0.0
        assert e_split <= max(1.5 * e_f32, ABS_FLOOR), (z, e_split, e_f32)


def test_first_layer_takes_the_split_route_and_matches_the_fp32_route(monkeypatch):
    """one training-layout batch through the stack on both routes: outputs and every gradient agree to fp32 rounding, and the
    route notes say which product ran where"""
    import golden_cases as gc
    from taxoexpan_amd import TaxoExpan, ops
    from taxoexpan_amd.graph import BatchedDGLGraph
    spec = gc.CASES["mag_pgat_wmr_lbm_q8x32"]
    shapes, x, q = gc.make_inputs(spec)
    params = gc.make_params(spec)
    res = {}
    for route in ("bf16x6", "fp32"):
        monkeypatch.setattr(ops, "_NO_SPLIT_GEMM", route == "fp32")
        model = TaxoExpan("PGAT", "WMR", "LBM", in_dim=250, hidden_dim=500, out_dim=500, pos_dim=50, num_layers=1, heads=[4, 1],
                          feat_drop=0.0, attn_drop=0.0, hidden_drop=0.0, out_drop=0.0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        model = model.to(_dev()).train()
        g = BatchedDGLGraph.from_egonet_shapes([s[0] for s in shapes], [s[1] for s in shapes])
        scores = model(g, torch.from_numpy(x).to(_dev()), torch.from_numpy(q).to(_dev()))
        assert ops.ROUTES.get("proj") == route
        scores.reshape(spec["n_queries"], -1).logsumexp(1).sum().backward()
        torch.cuda.synchronize()
        res[route] = (scores.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in model.named_parameters() if p.grad is not None})
    np.testing.assert_allclose(res["bf16x6"][0], res["fp32"][0], rtol=1e-4, atol=2e-5)
    for k, gref in res["fp32"][1].items():
        np.testing.assert_allclose(res["bf16x6"][1][k], gref, rtol=2e-3, atol=2e-4 * float(np.abs(gref).max()) + 1e-12, err_msg=k)


@pytest.mark.parametrize("N,Kh,Pd,H,D,p", [(1000, 2000, 50, 4, 500, 0.1), (300, 64, 16, 2, 63, 0.25), (129, 250, 50, 1, 126, 0.0)])
def test_input_gradient_of_a_deeper_layer_on_the_split_route(N, Kh, Pd, H, D, p):
    """txe_gat_dense_bwd with need_dh = 1 (a layer above the first: d_X = d_Y Wp over ALL columns, dropout mask and leaky' factor in
    the epilogue): with the extra workspace the product runs on the bf16 pipe (d_Y packed as rows, Wp packed from its transpose) --
    against float64 and against the fp32-MFMA route of the same call"""
    import ctypes
    from taxoexpan_amd import _lib
    dev = _dev()
    rs = np.random.RandomState(5 + N)
    vocab = 3
    Kt, F = Kh + Pd, H * D
    Kp, Fp = _lib.call("txe_gat_padded_k", Kh, Pd), _lib.call("txe_gat_padded_f", H, D)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    W, al, ar = f32(rs.standard_normal((F, Kt)) * 0.1), f32(rs.standard_normal(F)), f32(rs.standard_normal(F))
    Wp = torch.zeros((Fp, Kp), device=dev)
    _lib.call("txe_gat_pack_weights", W.data_ptr(), al.data_ptr(), ar.data_ptr(), H, D, Kt, Wp.data_ptr(), _lib.stream_ptr())
    X = torch.zeros((N, Kp), device=dev)
    X[:, :Kt] = f32(rs.standard_normal((N, Kt)))
    pos = torch.from_numpy(rs.randint(0, vocab, N).astype(np.int32)).to(dev)
    dY = torch.zeros((N, Fp), device=dev)
    dY[:, :F + 2 * H] = f32(rs.standard_normal((N, F + 2 * H)))
    mask = None
    if p > 0:
        mask = torch.empty((N, (Kt + 31) // 32), dtype=torch.int32, device=dev)
        _lib.call("txe_dropout_mask", N, Kt, p, 99, mask.data_ptr(), _lib.stream_ptr())
    base = _lib.call("txe_gat_dense_ws_bytes", N, Kh, Pd, H, D, vocab)
    extra = _lib.call("txe_gat_dense_bwd_split_ws_bytes", N, Kh, Pd, H, D)
    assert extra > 0
    slope = 0.2
    outs = []
    for wsb in (base + extra, base):
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        dX = torch.full((N, Kp), float("nan"), device=dev)
        dW, dal, dar, dP = torch.empty_like(W), torch.empty_like(al), torch.empty_like(ar), torch.empty((vocab, Pd), device=dev)
        _lib.call("txe_gat_dense_bwd", X.data_ptr(), N, Kh, Pd, pos.data_ptr(), vocab, Wp.data_ptr(), W.data_ptr(), al.data_ptr(), ar.data_ptr(),
                  H, D, p, mask.data_ptr() if mask is not None else None, dY.data_ptr(), 1, 1, slope, dX.data_ptr(), dW.data_ptr(), dal.data_ptr(),
                  dar.data_ptr(), dP.data_ptr(), 0, None, 7, None, ws.data_ptr(), wsb, _lib.stream_ptr())
        torch.cuda.synchronize()
        assert torch.isnan(dX[:, Kt:]).all() or Kt == Kp                # padding columns are not this call's to write
        outs.append((dX[:, :Kt].cpu().double().numpy(), dW.cpu().double().numpy(), dP.cpu().double().numpy()))
    keep = np.ones((N, Kt))
    if mask is not None:
        bits = ((mask.cpu().numpy().astype(np.int64)[:, :, None] >> np.arange(32)) & 1).reshape(N, -1)[:, :Kt]
        keep = bits / (1.0 - p)
    Wd, dYd, Xd = W.cpu().double().numpy(), dY.cpu().double().numpy(), X.cpu().double().numpy()[:, :Kt]
    ald, ard = al.cpu().double().numpy(), ar.cpu().double().numpy()
    wa = np.stack([(ald.reshape(H, D)[h][:, None] * Wd[h * D:(h + 1) * D]).sum(0) for h in range(H)] +
                  [(ard.reshape(H, D)[h][:, None] * Wd[h * D:(h + 1) * D]).sum(0) for h in range(H)])
    ref = (dYd[:, :F + 2 * H] @ np.concatenate([Wd, wa], axis=0)) * keep
    ref[:, :Kh] *= np.where(Xd[:, :Kh] > 0, 1.0, slope)
    scale = np.abs(ref).max()
    for dX, _, _ in outs:
        assert np.abs(dX - ref).max() <= 2e-5 * scale
    assert np.abs(outs[0][0] - outs[1][0]).max() <= 4e-6 * scale            # the two routes: fp32 rounding apart
    np.testing.assert_array_equal(outs[0][1], outs[1][1])                    # (dW does not depend on the d_X route)


def test_seeded_shape_fuzz_of_the_three_products():
    """60 random shapes each of the NT product (ragged M / N / K, odd leading dimensions), the TN product (any 4-aligned width, slices that
    end past the last row) and the plain-GEMM route bit: against float64, error no worse than 1.5 x an fp32 product's"""
    from taxoexpan_amd import _lib
    rs = np.random.RandomState(2024)
    s = _lib.stream_ptr()
    dev = _dev()
    for it in range(60):
        M, N, K = int(rs.randint(1, 700)), int(rs.randint(1, 500)), int(rs.randint(1, 400))
        lda, ldb, ldc = K + int(rs.randint(0, 5)), K + int(rs.randint(0, 5)), N + int(rs.randint(0, 7))
        A = torch.randn(M, lda, device=dev)[:, :K]
        B = torch.randn(N, ldb, device=dev)[:, :K]
        C = torch.full((M, ldc), float("nan"), device=dev)
        wsb = _lib.call("txe_gemm_plain_split_ws_bytes", M, N, K)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.call("txe_gemm_plain", 0, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc, M, N, K, 1, 8, ws.data_ptr(), wsb, s)
        ref = A.double() @ B.double().t()
        scale = (A.double().abs() @ B.double().abs().t()).clamp_min(1e-300)
        e_split, e_f32 = _err(C[:, :N], ref, scale), _err(A @ B.t(), ref, scale)
        assert torch.isnan(C[:, N:]).all() and e_split <= max(1.5 * e_f32, ABS_FLOOR), (it, M, N, K, e_split, e_f32)
    for it in range(60):
        n, M, N, S = int(rs.randint(1, 3000)), 128 * int(rs.randint(1, 4)), 4 * int(rs.randint(1, 120)), int(rs.randint(1, 9))
        A = (torch.randn(n, M, device=dev) * 0.1)
        B = torch.randn(n, N, device=dev)
        Bt = torch.empty(_lib.call("txe_split_packed_t_bytes", n, N), dtype=torch.uint8, device=dev)
        ks = (((n + S - 1) // S) + 15) // 16 * 16
        part = torch.full((S, M, N), float("nan"), device=dev)
        _lib.call("txe_split_pack_t", B.data_ptr(), N, n, N, Bt.data_ptr(), s)
        _lib.call("txe_gemm_tn_split", A.data_ptr(), M, M, Bt.data_ptr(), N, n, S, ks, part.data_ptr(), N, M * N, s)
        assert torch.isfinite(part).all(), (it, n, M, N, S)
        ref = A.double().t() @ B.double()
        scale = (A.double().abs().t() @ B.double().abs()).clamp_min(1e-300)
        e_split, e_f32 = _err(part.double().sum(0), ref, scale), _err(A.t() @ B, ref, scale)
        assert e_split <= max(1.5 * e_f32, ABS_FLOOR), (it, n, M, N, S, e_split, e_f32)


def test_bilinear_projection_on_the_split_route():
    """U = hg W (the scoring loop's factored half) with the scratch: against float64 and the fp32-MFMA route"""
    from taxoexpan_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    G, l, r = 3001, 500, 256
    hg, W = torch.randn(G, l, generator=g).to(dev), (torch.randn(l, r, generator=g) * 0.05).to(dev)
    outs = []
    for split in (True, False):
        U = torch.full((G, r), float("nan"), device=dev)
        wsb = _lib.call("txe_gemm_plain_split_ws_bytes", G, r, l) if split else 0
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        _lib.call("txe_bilinear_project", hg.data_ptr(), l, G, l, W.data_ptr(), r, U.data_ptr(), r, ws.data_ptr() if split else None, wsb, _lib.stream_ptr())
        torch.cuda.synchronize()
        outs.append(U)
    ref = hg.double() @ W.double()
    scale = (hg.double().abs() @ W.double().abs()).clamp_min(1e-300)
    e_split, e_f32 = _err(outs[0], ref, scale), _err(outs[1], ref, scale)
    assert e_split <= max(1.5 * e_f32, ABS_FLOOR), (e_split, e_f32)
