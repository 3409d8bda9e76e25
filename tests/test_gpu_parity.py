"""GPU parity tests proper: the HIP path (through the C ABI, include/txe.h) against (a) the golden vectors captured from
the unmodified reference and (b) the CPU oracle on the same seeded inputs.  Tolerance: BASELINE.json's north star asks
logits / ranking within 1e-4 fp32.  Model-level gradients are gated against the oracle run in FLOAT64 with the unmodified reference's
own fp32 gradient (the golden) as the yardstick: max |HIP - f64| <= max(2 x max |reference fp32 - f64|, 2e-5 max |f64|) and
<= 1e-4 max |f64| per tensor unless fp32 itself cannot (golden_util.gate_against_f64) -- and, entry by entry, 2e-3 relative against the golden itself."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_cases as gc
import txe_oracle as orc
from golden_util import GOLDEN_DIR, check_grad, gate_against_f64, golden_grad_entries, gradient_scale_floor, load_case, oracle_gradients_f64

pytestmark = pytest.mark.gpu

RT, AT = 1e-4, 2e-5


def _dev():
    return torch.device("cuda:0")


def _build_model(spec, params):
    from taxoexpan_amd import TaxoExpan
    drop = spec.get("dropout")
    opts = dict(in_dim=spec["in_dim"], hidden_dim=spec["hidden_dim"], out_dim=spec["out_dim"], pos_dim=spec["pos_dim"],
                num_layers=spec["num_layers"], heads=spec["heads"], feat_drop=(drop[0] if drop else 0.1),
                attn_drop=(drop[1] if drop else 0.1), hidden_drop=(drop[0] if drop else 0.1), out_drop=(drop[0] if drop else 0.1))
    model = TaxoExpan(spec["prop"], spec["readout"], spec["match"], **opts)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)   # same names/shapes as the reference
    return model.to(_dev())


def _graph(shapes):
    from taxoexpan_amd.graph import BatchedDGLGraph
    return BatchedDGLGraph.from_egonet_shapes([s[0] for s in shapes], [s[1] for s in shapes])


NODROP = [n for n, s in gc.CASES.items() if not s.get("dropout")]       # includes the ConcatReadout + MLP case


def _values(o):
    """the numbers of a readout's return value -- a tensor or a model_zoo.DeferredGraphVector -- read AFTER the step, outside autograd"""
    with torch.no_grad():
        return (o if torch.is_tensor(o) else o.tensor()).detach().cpu().numpy()


@pytest.mark.parametrize("name", NODROP)
def test_model_matches_reference_goldens(name):
    """every dropout-free golden of the unmodified reference (oracle/gen_golden.py): node states, graph vectors, scores, loss, gradients.
    The `*_q8x32` cases are training batches in the trainer's layout (256 egonets, each query row stacked 32 times): there the step must
    take the graph vector FOLDED into the matcher, with the matcher's query-side job riding in the stack's Z sweep -- the route bench.py
    times, pinned here to the reference itself.  Nothing touches the readout's return value before the matcher has consumed it."""
    from taxoexpan_amd import model_zoo as mz, ops
    spec, z, shapes, x, q, params, graph = load_case(name)
    model = _build_model(spec, params).eval()
    g = _graph(shapes)
    caps = {}
    model.readout.register_forward_hook(lambda m, i, o: caps.__setitem__("hg", o))       # (the object: looked at after the step)
    with ops.debug_capture() as runs:
        scores = model(g, torch.from_numpy(x).to(_dev()), torch.from_numpy(q).to(_dev()))
    nq = spec["n_queries"]
    loss = torch.nn.functional.cross_entropy(scores.reshape(nq, -1), torch.zeros(nq, dtype=torch.long, device=_dev()), reduction="sum")
    loss.backward()
    if spec.get("repeat_queries") and not (ops._NO_MATCH_FOLD or ops._NO_FUSED_BWD or ops._NO_QUERY_RUNS or mz._NO_FOLD):
        edot = not ops._NO_FOLD_EDOT
        taken = dict(runs.routes)
        assert (taken["match"], taken["stack"], taken["fold"]) == ("folded", "collapse_z" + ("+edot" if edot else ""), "edot" if edot else "inline"), taken
        assert ops.ROUTES["stack_bwd"] == ("fused+edot" if edot else "collapse")
    step = gc.row_steps(spec)[0]
    np.testing.assert_allclose(g.ndata["h"].detach().cpu().numpy()[::step], z["hn"], rtol=RT, atol=AT)
    np.testing.assert_allclose(_values(caps["hg"]), z["hg"], rtol=RT, atol=AT)
    np.testing.assert_allclose(scores.detach().cpu().numpy(), z["scores"], rtol=RT, atol=AT)
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-4)
    # gradients: reference = the float64 oracle on the same inputs; yardstick = the UNMODIFIED REFERENCE's own fp32 gradient (the golden)
    for k, p in model.named_parameters():
        check_grad(z, k, p.grad.cpu().numpy(), rtol=2e-3, atol=2e-5)      # entry by entry against the golden
    if spec["match"] == "MLP":                                            # (the oracle's model-level forward has the two bilinear matchers)
        return
    g64, _ = oracle_gradients_f64(spec, params, graph, x, q)
    errors, report, sf = [], [], gradient_scale_floor(g64)
    for k, p in model.named_parameters():
        gold, (got, ref) = golden_grad_entries(z, k, p.grad.cpu().numpy(), g64[k])
        gate_against_f64(got, ref, gold, "grad " + k, errors, report, scale_floor=sf)
    print(f"\n[{name}] worst gradient error vs float64, fraction of max |ref|: HIP {max(r[1] for r in report):.2e}, "
          f"reference fp32 {max(r[2] for r in report):.2e}")
    assert not errors, "\n".join(errors)


def _hash_masks(spec, params, graph, seed, csr_eid_in):
    """the masks the kernels will regenerate from `seed`, in the oracle's kwargs form"""
    from taxoexpan_amd import rng
    pf, pa = spec["dropout"]
    is_gat = spec["prop"] in ("PGAT", "GAT")
    N, E = graph["num_nodes"], int(graph["src"].numel())
    out = []
    for l in range(spec["num_layers"] + 1):
        if is_gat:
            kt = params[f"graph_propagate.gat_layers.{l}.fc.weight"].shape[1]
            H = spec["heads"][l]
            d = dict(feat_keep=torch.from_numpy(rng.keep_mask_bits(seed + 16 * l, N, kt, pf)), feat_scale=1.0 / (1.0 - pf))
            if pa > 0:
                m_csr = rng.keep_mask(seed + 16 * l + 1, (E, H), pa)       # destination-CSR order
                m_eid = np.empty_like(m_csr)
                m_eid[csr_eid_in] = m_csr
                d.update(attn_keep=torch.from_numpy(m_eid).unsqueeze(-1), attn_scale=1.0 / (1.0 - pa))
            out.append(d)
        else:
            kt = params[f"graph_propagate.layers.{l}.weight"].shape[0]
            out.append(dict(keep=torch.from_numpy(rng.keep_mask_bits(seed + 16 * l, N, kt, pf)), keep_scale=1.0 / (1.0 - pf)))
    return out


@pytest.mark.parametrize("name,drop", [("small_pgat_dropout", None), ("small_pgcn_dropout", None),
                                       # the inputs / parameters of dropout-free cases with dropout switched on: a middle layer (its
                                       # input dropped by the aggregation below), an output layer that is not folded (2 heads)
                                       ("small_pgat_2layer", (0.3, 0.25)), ("small_gat_mr_bim", (0.2, 0.1))])
def test_training_mode_dropout_matches_oracle(name, drop, monkeypatch):
    from taxoexpan_amd import ops
    spec, z, shapes, x, q, params, graph = load_case(name)
    if drop is not None:
        spec = dict(spec, dropout=drop)
    seed = 123456789
    monkeypatch.setattr(ops, "new_seed", lambda: seed)
    model = _build_model(spec, params).train()
    g = _graph(shapes)
    eid_in = g.csr("cpu").eid_in.numpy()
    xg = torch.from_numpy(x).to(_dev()).requires_grad_(True)       # (the gradient to the node features too: the first layer's input is
    scores = model(g, xg, torch.from_numpy(q).to(_dev()))          #  stored dropped, its d_X comes out of the masked GEMM epilogue)
    nq = spec["n_queries"]
    loss = torch.nn.functional.cross_entropy(scores.reshape(nq, -1), torch.zeros(nq, dtype=torch.long, device=_dev()), reduction="sum")
    loss.backward()
    P = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    masks = _hash_masks(spec, params, graph, seed, eid_in)
    xc = torch.from_numpy(x).clone().requires_grad_(True)
    s_ref, hg_ref, hn_ref = orc.taxoexpan_forward(P, graph, xc, torch.from_numpy(q), spec["prop"], spec["readout"],
                                                  spec["match"], spec["heads"], spec["num_layers"], masks)
    l_ref = orc.info_nce_loss(s_ref, nq)
    l_ref.backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=2e-3, atol=2e-5, err_msg="d node features")
    np.testing.assert_allclose(g.ndata["h"].detach().cpu().numpy(), hn_ref.detach().numpy(), rtol=RT, atol=AT)
    np.testing.assert_allclose(scores.detach().cpu().numpy(), s_ref.detach().numpy(), rtol=RT, atol=AT)
    g64, dx64 = oracle_gradients_f64(spec, params, graph, x, q, masks, with_x=True)      # the reference: float64; the yardstick: the fp32 oracle
    errors, sf = [], gradient_scale_floor(g64)
    gate_against_f64(xg.grad.cpu().numpy(), dx64, xc.grad.numpy(), "d node features", errors)
    for k, p in model.named_parameters():
        gate_against_f64(p.grad.cpu().numpy(), g64[k], P[k].grad.numpy(), "grad " + k, errors, scale_floor=sf)
        np.testing.assert_allclose(p.grad.cpu().numpy(), P[k].grad.numpy(), rtol=2e-3, atol=2e-5, err_msg=k)
    assert not errors, "\n".join(errors)


@pytest.mark.parametrize("name", ["mag_pgat_wmr_lbm_q8x32", "semeval_pgat_wmr_bim_q8x32"])
def test_forward_egonet_walk_matches_reference_goldens(name, monkeypatch):
    """the four-head goldens of the unmodified reference with the forward message/reduce sweep FORCED onto the egonet walk
    (gat_aggregate_ego_kernel; by itself it takes batches of 4,096 nodes and more -- the full-size tests): node states, graph vectors,
    scores, loss, every gradient entry"""
    from taxoexpan_amd import ops
    monkeypatch.setattr(ops, "_FWD_SWEEP", 3)
    spec, z, shapes, x, q, params, graph = load_case(name)
    model = _build_model(spec, params).eval()
    g = _graph(shapes)
    caps = {}
    model.readout.register_forward_hook(lambda m, i, o: caps.__setitem__("hg", o))
    scores = model(g, torch.from_numpy(x).to(_dev()), torch.from_numpy(q).to(_dev()))
    nq = spec["n_queries"]
    loss = torch.nn.functional.cross_entropy(scores.reshape(nq, -1), torch.zeros(nq, dtype=torch.long, device=_dev()), reduction="sum")
    loss.backward()
    step = gc.row_steps(spec)[0]
    np.testing.assert_allclose(g.ndata["h"].detach().cpu().numpy()[::step], z["hn"], rtol=RT, atol=AT)
    np.testing.assert_allclose(_values(caps["hg"]), z["hg"], rtol=RT, atol=AT)
    np.testing.assert_allclose(scores.detach().cpu().numpy(), z["scores"], rtol=RT, atol=AT)
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-4)
    for k, p in model.named_parameters():
        check_grad(z, k, p.grad.cpu().numpy(), rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("switch,case", [("_NO_SIDE_STREAM", "small_pgat_2layer"), ("_NO_FUSED_BWD", "small_pgat_2layer"),
                                         ("_NO_FUSED_LOGITS", "small_pgat_2layer"), ("_NO_TAIL_CHAIN", "small_pgat_2layer"),
                                         # four heads under the folded layer: the egonet-walking sweep (windows cut the larger egonets:
                                         # foreign hubs, hpart rows, the fix-up pass) against the per-out-edge sweep
                                         ("_NO_EGO_WALK", "mag_pgat_wmr_lbm_q8x32"), ("_NO_EGO_WALK", "semeval_pgat_wmr_bim_q8x32"),
                                         # the backward walk staged from the batch's plan against staging from the CSR arrays
                                         ("_NO_WALK_PLAN", "mag_pgat_wmr_lbm_q8x32"), ("_NO_WALK_PLAN", "semeval_pgat_wmr_bim_q8x32"),
                                         # the first layer's input never stored (the packs form it) against the stored one
                                         ("_NO_VIRTUAL_X", "mag_pgat_wmr_lbm_q8x32"), ("_NO_VIRTUAL_X", "small_pgat_2layer"),
                                         ("_NO_VIRTUAL_X", "semeval_pgat_wmr_bim_q8x32")])
def test_ab_switch_routes_give_the_same_training_step(switch, case, monkeypatch):
    """every route attribute of ops.py selects another ROUTE to the same numbers: a training step (dropout on, fixed seed) with the switch
    set against the default -- scores and every gradient (the per-layer preparation entry once left the stored-dropped flag of its
    argument block uninitialised: the default route never noticed)"""
    from taxoexpan_amd import ops
    spec, z, shapes, x, q, params, graph = load_case(case)
    spec = dict(spec, dropout=(0.3, 0.25))
    monkeypatch.setattr(ops, "new_seed", lambda: 424242)
    outs = []
    if switch == "_NO_EGO_WALK":       # (these batches are below the size at which the forward sweep walks egonets by itself)
        monkeypatch.setattr(ops, "_FWD_SWEEP", 3)
    for on in (False, True):
        monkeypatch.setattr(ops, switch, on)
        model = _build_model(spec, params).train()
        xg = torch.from_numpy(x).to(_dev()).requires_grad_(True)
        s = model(_graph(shapes), xg, torch.from_numpy(q).to(_dev()))
        nq = spec["n_queries"]
        torch.nn.functional.cross_entropy(s.reshape(nq, -1), torch.zeros(nq, dtype=torch.long, device=_dev()), reduction="sum").backward()
        torch.cuda.synchronize()
        outs.append((s.detach().cpu().numpy(), xg.grad.cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}))
    assert np.isfinite(outs[1][0]).all()
    if switch == "_NO_VIRTUAL_X":      # the same planes reach the same products: bit for bit
        assert np.array_equal(outs[1][0], outs[0][0]) and np.array_equal(outs[1][1], outs[0][1])
        assert all(np.array_equal(outs[1][2][k], outs[0][2][k]) for k in outs[0][2])
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=1e-4, atol=1e-6)
    for k in outs[0][2]:
        np.testing.assert_allclose(outs[1][2][k], outs[0][2][k], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("N,Kh,ld_h,Pd,p,from_h", [(517, 250, 250, 50, 0.3, True), (517, 250, 250, 50, 0.5, True), (301, 251, 253, 50, 0.1, True),
                                                     (64, 96, 96, 0, 0.5, True), (33, 7, 8, 3, 0.25, True), (129, 2000, 0, 50, 0.5, False),
                                                     (77, 250, 0, 50, 0.1, False), (5, 16500, 16500, 50, 0.5, True), (1, 250, 250, 50, 0.5, True)])
def test_prepare_entries_agree_and_stored_dropped_input_matches_its_mask(N, Kh, ld_h, Pd, p, from_h):
    """txe_gat_layer_prepare (one layer) == txe_gat_layers_prepare (the stack's one launch) on X / Wp / mask; with x_dropped the stored
    input is the plain one times keep(mask) / (1 - p), from the very bits written into the mask (rng.keep_mask_bits restates them).
    Geometries: 8-byte and 4-byte feature rows, no position columns, rows narrower than a quad, feature columns written by a producer
    (h == NULL: only the position / padding columns are touched, the quad straddling Kh keeps its feature part), rows wider than one
    pass of a workgroup, a single row; p = 0.5 (one bit plane) and thresholds with many planes"""
    import ctypes
    from taxoexpan_amd import _lib, rng
    dev = _dev()
    rs = np.random.RandomState(3 + Kh)
    H, D, seed = 2, 12, 9876543210
    Kt = Kh + Pd
    Kp, Fp = _lib.call("txe_gat_padded_k", Kh, Pd), _lib.call("txe_gat_padded_f", H, D)
    hfull = torch.from_numpy(rs.standard_normal((max(N, 1), max(ld_h, Kh))).astype(np.float32)).to(dev)
    h = hfull[:N, :Kh]
    pos_full = torch.from_numpy(rs.randint(0, 3, max(N, 1)).astype(np.int32)).to(dev)
    pos = pos_full[:N]
    P = torch.from_numpy(rs.standard_normal((3, max(Pd, 1))).astype(np.float32)).to(dev)[:, :Pd].contiguous() if Pd else None
    W = torch.from_numpy(rs.standard_normal((H * D, Kt)).astype(np.float32)).to(dev)
    al, ar = (torch.from_numpy(rs.standard_normal((1, H, D)).astype(np.float32)).to(dev) for _ in range(2))
    wpr = (Kt + 31) // 32
    pre = torch.from_numpy(rs.standard_normal((max(N, 1), Kp)).astype(np.float32)).to(dev)[:N]     # what a producer left in X (h == NULL)
    ptr = lambda t: t.data_ptr() if t is not None else None

    def bufs():
        X = torch.full((max(N, 1), Kp), 7.0, device=dev)[:N] if from_h else pre.clone()
        return X, torch.full((Fp, Kp), 7.0, device=dev), torch.zeros((max(N, 1), wpr), dtype=torch.int32, device=dev)[:N]

    def multi(x_dropped):
        X, Wp, mask = bufs()
        d = (_lib.GatPrepareDesc * 1)()
        d[0].h, d[0].ld_h, d[0].n_nodes, d[0].Kh, d[0].pos, d[0].P, d[0].Pd = (hfull.data_ptr() if from_h else None), ld_h, N, Kh, ptr(pos_full), ptr(P), Pd
        d[0].X, d[0].W, d[0].attn_l, d[0].attn_r, d[0].H, d[0].D, d[0].Wp = X.data_ptr(), W.data_ptr(), al.data_ptr(), ar.data_ptr(), H, D, Wp.data_ptr()
        d[0].feat_drop_p, d[0].seed, d[0].mask, d[0].x_dropped = p, seed, mask.data_ptr(), x_dropped
        _lib.call("txe_gat_layers_prepare", ctypes.cast(d, ctypes.c_void_p), 1, _lib.stream_ptr())
        torch.cuda.synchronize()
        return X, Wp, mask
    X1, Wp1, m1 = bufs()
    _lib.call("txe_gat_layer_prepare", hfull.data_ptr() if from_h else None, ld_h, N, Kh, ptr(pos_full), ptr(P), Pd, X1.data_ptr(), W.data_ptr(),
              al.data_ptr(), ar.data_ptr(), H, D, Wp1.data_ptr(), p, seed, m1.data_ptr(), _lib.stream_ptr())
    torch.cuda.synchronize()
    X2, Wp2, m2 = multi(0)
    assert torch.equal(X1, X2) and torch.equal(Wp1, Wp2) and torch.equal(m1, m2)
    feat = h if from_h else pre[:, :Kh]
    cols = [feat] + ([P[pos.long()]] if Pd else []) + [torch.zeros(N, Kp - Kt, device=dev)]
    want = torch.cat(cols, dim=1)
    assert torch.equal(X2, want)
    assert torch.equal(Wp2[:H * D, :Kt], W) and float(Wp2[:H * D, Kt:].abs().max() if Kp > Kt else 0.0) == 0.0
    keep = torch.from_numpy(rng.keep_mask_bits(seed, N, Kt, p)).to(dev)
    bits = ((m2.long().unsqueeze(-1) >> torch.arange(32, device=dev)) & 1).reshape(N, wpr * 32)[:, :Kt].float()
    assert torch.equal(bits, keep)                                   # the mask job's words
    X3, Wp3, m3 = multi(1)
    assert torch.equal(Wp3, Wp2) and torch.equal(m3, m2)             # (with h the words come from the build job itself)
    want3 = want.clone()
    c0 = 0 if from_h else Kh                                         # (h == NULL: the producer's columns are not touched)
    want3[:, c0:Kt] = want[:, c0:Kt] * keep[:, c0:] * (1.0 / (1.0 - p))
    assert torch.equal(X3, want3)


@pytest.mark.parametrize("N,Kh,Pd,H,D,vocab,p", [(1000, 250, 50, 4, 500, 3, 0.1), (517, 300, 50, 4, 600, 3, 0.5), (33, 250, 50, 1, 126, 3, 0.0),
                                                  (16, 64, 16, 2, 63, 8, 0.25), (1, 250, 50, 4, 500, 3, 0.1), (4099, 7, 3, 3, 40, 5, 0.3),
                                                  (200, 250, 62, 2, 64, 2, 0.2)])
def test_first_layer_position_dx_streaming_kernel(N, Kh, Pd, H, D, vocab, p):
    """txe_gat_dense_bwd with need_dh = 0 (a first PGAT layer, model_zoo.py:214-215 backward): the streaming position-column kernel
    (txe_dxpos.hip: d_X[:, c0:Kt] and dP in one pass over d_Y) against float64 -- d_X position columns, dP, and dW / d_attn through
    the split-K slices + the chained phase B (once launched in place, once deferred into a chain and flushed: bit-equal).  Shapes: MAG and SemEval dimensions (the SemEval slab runs past the padded row: clamped
    column vectors), rows that are no multiple of the 16-row workgroups, a single row, 8 position classes, odd widths, no dropout."""
    from taxoexpan_amd import _lib
    dev = _dev()
    assert _lib.call("txe_gat_dx_streams", Kh, Pd, 0) == 1 and _lib.call("txe_gat_dx_streams", Kh, Pd, 1) == 0
    rs = np.random.RandomState(11 + N)
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
        _lib.call("txe_dropout_mask", N, Kt, p, 777, mask.data_ptr(), _lib.stream_ptr())
    wsb = _lib.call("txe_gat_dense_ws_bytes", N, Kh, Pd, H, D, vocab)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    import ctypes
    outs = []
    for deferred in (False, True):
        dX = torch.full((N, Kp), float("nan"), device=dev)
        dW, dal, dar, dP = torch.empty_like(W), torch.empty_like(al), torch.empty_like(ar), torch.empty((vocab, Pd), device=dev)
        chain = ctypes.create_string_buffer(_lib.TAIL_CHAIN_BYTES)
        cp = ctypes.cast(chain, ctypes.c_void_p)
        _lib.call("txe_gat_dense_bwd", X.data_ptr(), N, Kh, Pd, pos.data_ptr(), vocab, Wp.data_ptr(), W.data_ptr(), al.data_ptr(), ar.data_ptr(),
                  H, D, p, mask.data_ptr() if mask is not None else None, dY.data_ptr(), 0, 0, 1.0, dX.data_ptr(), dW.data_ptr(), dal.data_ptr(),
                  dar.data_ptr(), dP.data_ptr(), 0, None, 7 | (64 if deferred else 0), cp if deferred else None, ws.data_ptr(), wsb, _lib.stream_ptr())
        if deferred:
            _lib.call("txe_gat_tail_flush", cp, _lib.stream_ptr())
        torch.cuda.synchronize()
        outs.append([t.cpu().double().numpy() for t in (dX[:, Kh:Kt], dW, dal, dar, dP)])
    # float64 restatement
    keep = np.ones((N, Kt))
    if mask is not None:
        bits = ((mask.cpu().numpy().astype(np.int64)[:, :, None] >> np.arange(32)) & 1).reshape(N, -1)[:, :Kt]
        keep = bits / (1.0 - p)
    Wd, dYd, Xd = W.cpu().double().numpy(), dY.cpu().double().numpy(), X.cpu().double().numpy()[:, :Kt]
    ald, ard = al.cpu().double().numpy(), ar.cpu().double().numpy()
    wa = np.stack([(ald.reshape(H, D)[h][:, None] * Wd[h * D:(h + 1) * D]).sum(0) for h in range(H)] +
                  [(ard.reshape(H, D)[h][:, None] * Wd[h * D:(h + 1) * D]).sum(0) for h in range(H)])        # folded rows [2H][Kt]
    Wfull = np.concatenate([Wd, wa], axis=0)                                                                   # [F + 2H][Kt]
    dX_ref = (dYd[:, :F + 2 * H] @ Wfull) * keep
    dP_ref = np.stack([dX_ref[pos.cpu().numpy() == c][:, Kh:].sum(0) for c in range(vocab)])
    scale = np.abs(dX_ref[:, Kh:]).max() + 1e-30
    for o in outs:
        assert np.isfinite(o[0]).all()
        np.testing.assert_allclose(o[0], dX_ref[:, Kh:], rtol=1e-4, atol=1e-5 * scale)
        np.testing.assert_allclose(o[4], dP_ref, rtol=1e-4, atol=2e-5 * max(np.abs(dP_ref).max(), 1e-30))
    for a, b in zip(outs[0], outs[1]):                                # launched in place = deferred and flushed
        np.testing.assert_array_equal(a, b)
    dW_ref = (dYd[:, :F].T @ (Xd * keep)) + ald[:, None] * (dYd[:, F:F + H].T @ (Xd * keep)).repeat(D, 0) + ard[:, None] * (dYd[:, F + H:F + 2 * H].T @ (Xd * keep)).repeat(D, 0)
    np.testing.assert_allclose(outs[0][1], dW_ref, rtol=1e-4, atol=2e-5 * np.abs(dW_ref).max())


def _random_graph(n, e, seed, zero_in=True):
    """generic multigraph: a hub with in-degree > 64 (multi-chunk path), some nodes without in-edges"""
    rs = np.random.RandomState(seed)
    src = rs.randint(0, n, size=e)
    dst = rs.randint(1 if zero_in else 0, n, size=e)      # node 0 never a destination when zero_in
    hub = n // 2
    src = np.concatenate([src, rs.randint(0, n, size=150)])
    dst = np.concatenate([dst, np.full(150, hub)])
    return src.astype(np.int64), dst.astype(np.int64)


@pytest.mark.parametrize("H,D,K", [(3, 5, 7), (4, 8, 12), (1, 500, 64), (2, 66, 10), (6, 12, 9), (4, 600, 20), (2, 1100, 8)])
def test_gat_layer_generic_graph_fwd_bwd(H, D, K):
    """GATLayer on an arbitrary CSR (degree >> 64, zero in-degree, odd / even / x4 widths) vs the oracle."""
    from taxoexpan_amd.graph import DGLGraph
    from taxoexpan_amd.model_zoo import GATLayer
    n = 97
    src, dst = _random_graph(n, 400, seed=H * 100 + D)
    g = DGLGraph()
    g.add_nodes(n)
    g.add_edges(src, dst)
    torch.manual_seed(0)
    layer = GATLayer(K, D, H, feat_drop=0.0, attn_drop=0.0).to(_dev())
    x = torch.randn(n, K)
    xg = x.to(_dev()).requires_grad_(True)
    out = layer(g, xg)
    wgt = torch.randn(n, H, D)
    (out * wgt.to(_dev())).sum().backward()
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    xc = x.clone().requires_grad_(True)
    ref = orc.gat_layer(torch.from_numpy(src), torch.from_numpy(dst), n, xc, P["fc.weight"], P["attn_l"], P["attn_r"], 0.2)
    (ref * wgt).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=RT, atol=AT)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(layer.fc.weight.grad.cpu().numpy(), P["fc.weight"].grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(layer.attn_l.grad.cpu().numpy(), P["attn_l"].grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(layer.attn_r.grad.cpu().numpy(), P["attn_r"].grad.numpy(), rtol=2e-3, atol=2e-5)
    assert torch.all(out[0] == 0)            # zero in-degree -> zero row (DGL sum semantics)


@pytest.mark.parametrize("K,Fo", [(7, 5), (64, 500), (13, 66)])
def test_gcn_layer_generic_graph_fwd_bwd(K, Fo):
    import torch.nn.functional as F
    from taxoexpan_amd.graph import DGLGraph
    from taxoexpan_amd.model_zoo import GCNLayer
    n = 83
    src, dst = _random_graph(n, 300, seed=K)
    g = DGLGraph()
    g.add_nodes(n)
    g.add_edges(src, dst)
    torch.manual_seed(1)
    layer = GCNLayer(K, Fo, F.leaky_relu, 0.0).to(_dev())
    x = torch.randn(n, K)
    xg = x.to(_dev()).requires_grad_(True)
    out = layer(g, xg)
    wgt = torch.randn(n, Fo)
    (out * wgt.to(_dev())).sum().backward()
    W = layer.weight.detach().cpu().clone().requires_grad_(True)
    b = layer.bias.detach().cpu().clone().requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    s, d = torch.from_numpy(src), torch.from_numpy(dst)
    ref = orc.gcn_layer(s, d, n, xc, W, b, orc.gcn_norm(d, n), act_slope=0.01)
    (ref * wgt).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=RT, atol=AT)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(layer.weight.grad.cpu().numpy(), W.grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(layer.bias.grad.cpu().numpy(), b.grad.numpy(), rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("prop", ["PGCN", "GCN", "PGAT"])
def test_propagation_with_another_activation_runs_layer_by_layer(prop):
    """model_zoo.py:128-137,155-167,210-220 with an activation the fused stack does not know (torch.tanh; model.py only ever passes
    F.leaky_relu): PGCN / GCN used to raise here, which changes a script's flow -- now the layers run one by one (the fused stack of
    one layer each, the activation and the concat by torch): outputs and gradients against the oracle's layers composed the same way"""
    from taxoexpan_amd import model_zoo as mz
    from taxoexpan_amd.graph import BatchedDGLGraph
    dev = _dev()
    rs = np.random.RandomState(5)
    k, m = rs.randint(1, 4, 9), rs.randint(0, 6, 9)
    g = BatchedDGLGraph.from_egonet_shapes(k, m)
    n = g.number_of_nodes()
    pos = g.ndata["pos"].clone()
    x = torch.randn(n, 12, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(2)
    if prop == "PGCN":
        model = mz.PGCN(12, 16, 8, 4, num_layers=1, activation=torch.tanh, in_dropout=0.0, hidden_dropout=0.0, output_dropout=0.0).to(dev)
    elif prop == "GCN":
        model = mz.GCN(12, 16, 8, num_layers=1, activation=torch.tanh, in_dropout=0.0, hidden_dropout=0.0, output_dropout=0.0).to(dev)
    else:
        model = mz.PGAT(12, 16, 8, 4, num_layers=1, heads=[2, 1], activation=torch.tanh, feat_drop=0.0, attn_drop=0.0).to(dev)
    xg = x.to(dev).requires_grad_(True)
    out = model(g, xg)
    out = out.tensor() if hasattr(out, "tensor") and not torch.is_tensor(out) else out
    wgt = torch.randn(n, 8, generator=torch.Generator().manual_seed(3))
    (out * wgt.to(dev)).sum().backward()
    P = {kk: v.detach().cpu().clone().requires_grad_(True) for kk, v in model.state_dict().items()}
    s_, d_ = torch.from_numpy(np.asarray(g._src)).long(), torch.from_numpy(np.asarray(g._dst)).long()
    xc = x.clone().requires_grad_(True)
    h = xc
    for l in range(2):
        emb = P.get(f"prop_position_embeddings.{l}.weight")
        hin = h if emb is None else torch.cat((h, emb[pos]), 1)
        if prop == "PGAT":
            h = orc.gat_layer(s_, d_, n, hin, P[f"gat_layers.{l}.fc.weight"], P[f"gat_layers.{l}.attn_l"], P[f"gat_layers.{l}.attn_r"])
            h = torch.tanh(h.flatten(1)) if l == 0 else h.mean(1)
        else:
            h = orc.gcn_layer(s_, d_, n, hin, P[f"layers.{l}.weight"], P[f"layers.{l}.bias"], orc.gcn_norm(d_, n), act_slope=None)
            h = torch.tanh(h) if l == 0 else h
    (h * wgt).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), h.detach().numpy(), rtol=RT, atol=AT)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=2e-3, atol=2e-5)
    for kk, p_ in model.named_parameters():
        np.testing.assert_allclose(p_.grad.cpu().numpy(), P[kk].grad.numpy(), rtol=2e-3, atol=2e-5, err_msg=kk)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (129, 257, 33), (300, 2008, 300), (77, 130, 2050), (256, 128, 64),
                                   (200, 160, 4000), (132, 480, 3000), (100, 2080, 1500)])
def test_gemm_layouts_against_fp64(M, N, K):
    """the three operand layouts of the MFMA GEMM through public entry points (bilinear project = NN, score block = NT,
    bilinear backward dW = TN) against an fp64 host product"""
    from taxoexpan_amd import ops
    rs = np.random.RandomState(M + N + K)
    a = rs.standard_normal((M, K)).astype(np.float32)
    b = rs.standard_normal((K, N)).astype(np.float32)
    A, B = torch.from_numpy(a).to(_dev()), torch.from_numpy(b).to(_dev())
    ref = a.astype(np.float64) @ b.astype(np.float64)
    tol = dict(rtol=1e-4, atol=1e-4 * np.sqrt(K))
    nn = ops.bilinear_project(A, B.unsqueeze(0)).cpu().numpy()                       # A[M,K] @ B[K,N]
    np.testing.assert_allclose(nn, ref, **tol)
    nt = ops.score_block(A, B.t().contiguous(), False).cpu().numpy()                   # A[M,K] @ (Bt[N,K])^T
    np.testing.assert_allclose(nt, ref, **tol)
    # TN: dW[l][r] = sum_i e1[i][l] * ds[i] e2[i][r]  with e1 = a^T-shaped operands
    e1 = torch.from_numpy(np.ascontiguousarray(a.T)).to(_dev()).requires_grad_(True)   # [K, M] -> G=K, l=M
    e2 = torch.from_numpy(b).to(_dev())                                                # [K, N] -> r=N
    W = torch.zeros(1, M, N, device=_dev(), requires_grad=True)
    s = ops.BilinearPairFunction.apply(e1, e2, W, False)
    s.sum().backward()
    np.testing.assert_allclose(W.grad[0].cpu().numpy(), ref, **tol)


@pytest.mark.parametrize("layout,M,N,K", [(0, 8300, 2048, 320), (1, 8300, 2040, 352), (2, 8200, 2048, 256), (0, 1024, 17000, 256),
                                          (0, 8300, 2048, 160)])
def test_gemm_whole_rounds_on_persistent_workgroups(layout, M, N, K):
    """gemm_persist_kernel (whole rounds of 128 x 128 tiles of a short-K plain product: C drained from registers under the next
    tile's k-loop) against the same product computed in pieces too small to qualify (gemm_kernel): bit-identical -- every layout,
    a ragged last row panel / column tile (left to gemm_kernel), an odd number of k-tiles (stage-buffer parity flips per tile), the
    4- and 8-k-tile drain schedules, the row-panel-fastest tile order; and against fp64"""
    from taxoexpan_amd import _lib
    dev = _dev()
    rs = np.random.RandomState(layout * 7 + K)
    ra, ca = (M, K) if layout < 2 else (K, M)
    rb, cb = (N, K) if layout == 0 else (K, N)
    ldb = -(-cb // 128) * 128 if layout else cb                                       # rows of B past N exist only where B is [N][K]
    rows_b = -(-rb // 128) * 128 if layout == 0 else rb
    A = torch.from_numpy(rs.standard_normal((ra, ca)).astype(np.float32)).to(dev)
    Bfull = torch.zeros(rows_b, ldb, device=dev)
    Bfull[:rb, :cb] = torch.from_numpy(rs.standard_normal((rb, cb)).astype(np.float32)).to(dev)
    ws = torch.empty(_lib.call("txe_gemm_tail_ws_bytes"), dtype=torch.uint8, device=dev)

    def run(a_ptr, lda, m, c):
        _lib.call("txe_gemm_plain", layout, a_ptr, lda, Bfull.data_ptr(), ldb, c.data_ptr(), N, m, N, K, 1, 0, ws.data_ptr(), ws.numel(),
                  _lib.stream_ptr())
    C = torch.full((M, N), 7.0, device=dev)
    run(A.data_ptr(), ca, M, C)
    torch.cuda.synchronize()
    pieces = []
    step = 3072 if N <= 4096 else 256                                                   # < 2 rounds of tiles per piece
    for m0 in range(0, M, step):
        m = min(step, M - m0)
        c = torch.empty(m, N, device=dev)
        a_ptr = A.data_ptr() + 4 * (m0 * ca if layout < 2 else m0)
        run(a_ptr, ca, m, c)
        pieces.append(c)
    Cp = torch.cat(pieces)
    # the whole rounds (2 x 512 tiles: 8,192 rows of 16 column tiles / 128 column tiles of 8 row panels) went to the persistent
    # kernel and accumulate in plain k order like the (unsplit) pieces; the tiles behind them are k-split by gemm_kernel's tail
    # splitting, i.e. summed in another order
    pr, pc = (M, 16384) if N > 4096 else (8192, N)
    assert torch.equal(C[:pr, :pc], Cp[:pr, :pc])
    np.testing.assert_allclose(C.cpu().numpy(), Cp.cpu().numpy(), rtol=1e-4, atol=1e-4 * np.sqrt(K))
    a = A.cpu().numpy().astype(np.float64)
    b = Bfull[:rb, :cb].cpu().numpy().astype(np.float64)
    ref = (a if layout < 2 else a.T) @ (b.T if layout == 0 else b)
    np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.sqrt(K))


@pytest.mark.parametrize("n_nodes", [17877, 9100, 8192])
def test_first_layer_projection_skips_the_zero_padded_k_columns(n_nodes):
    """txe_gat_dense_fwd on the MAG first layer's shape (K = 250 + 50 padded to 320, 2,008 output columns): the persistent kernel
    runs only the two eight-column groups of the last k-tile that hold data (Epi.k_valid; whole tiles AND the k-slices of the
    leftover tiles at 17,877 and 9,100 rows, the 8-k-tile drain schedule on the two whole rounds of 8,192) -- bit-identical to the same padded product through
    txe_gemm_plain, which multiplies the zero columns as well; and against fp64"""
    from taxoexpan_amd import _lib
    dev = _dev()
    Kh, Pd, H, D = 250, 50, 4, 500
    Kp, Fe = _lib.call("txe_gat_padded_k", Kh, Pd), H * D + 2 * H
    Fp = _lib.call("txe_gat_padded_f", H, D)
    rs = np.random.RandomState(n_nodes)
    X = torch.zeros(n_nodes, Kp, device=dev)
    X[:, :Kh + Pd] = torch.from_numpy(rs.standard_normal((n_nodes, Kh + Pd)).astype(np.float32)).to(dev)
    Wp = torch.zeros(Fp, Kp, device=dev)
    Wp[:Fe, :Kh + Pd] = torch.from_numpy(rs.standard_normal((Fe, Kh + Pd)).astype(np.float32)).to(dev)
    ws = torch.empty(max(_lib.call("txe_gemm_tail_ws_bytes"), _lib.call("txe_gat_dense_ws_bytes", n_nodes, Kh, Pd, H, D, 3)), dtype=torch.uint8,
                     device=dev)
    Y = torch.full((n_nodes, Fp), 3.0, device=dev)
    _lib.call("txe_gat_dense_fwd", X.data_ptr(), n_nodes, Kh, Pd, Wp.data_ptr(), H, D, 0.0, None, Y.data_ptr(), ws.data_ptr(), ws.numel(),
              _lib.stream_ptr())
    C = torch.full((n_nodes, Fp), 5.0, device=dev)
    _lib.call("txe_gemm_plain", 0, X.data_ptr(), Kp, Wp.data_ptr(), Kp, C.data_ptr(), Fp, n_nodes, Fe, Kp, 1, 0, ws.data_ptr(), ws.numel(),
              _lib.stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(Y[:, :Fe], C[:, :Fe])
    ref = X.double() @ Wp[:Fe].double().t()
    np.testing.assert_allclose(Y[:, :Fe].cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4 * np.sqrt(Kh + Pd))


def test_gemm_split_k_on_160_wide_tiles_against_fp64():
    """a big split-K TN product whose width 160-wide tiles cover with less padding (2080 = 13 x 160): the flat staging geometry and
    the register epilogue of gemm_kernel<false,false,4,4,160>, partial slices summed by the caller"""
    from taxoexpan_amd import _lib
    dev = _dev()
    M, N, K, S = 2048, 2080, 6016, 4
    rs = np.random.RandomState(9)
    A = torch.from_numpy(rs.standard_normal((K, M)).astype(np.float32)).to(dev)
    B = torch.from_numpy(rs.standard_normal((K, N)).astype(np.float32)).to(dev)
    C = torch.empty(S * M, N, device=dev)
    _lib.call("txe_gemm_plain", 2, A.data_ptr(), M, B.data_ptr(), N, C.data_ptr(), N, M, N, K, S, 0, None, 0, _lib.stream_ptr())
    got = C.view(S, M, N).double().sum(0).cpu().numpy()
    ref = A.cpu().numpy().astype(np.float64).T @ B.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.sqrt(K))


@pytest.mark.parametrize("M,N,K,S", [(2048, 320, 17877, 16), (1024, 2080, 6016, 4), (300, 316, 1000, 3), (128, 160, 40, 2), (132, 8, 33, 2),
                                     (2048, 320, 64, 2)])
def test_split_k_tn_product_with_lds_direct_copies(M, N, K, S):
    """gemm_tn_lds_kernel (txe_gemm_tnlds.h: operand tiles copied global -> LDS directly, permuted B fragments, vector stores from the
    accumulators) -- the first layer's weight-gradient shape with its ragged last k-tile per slice, widths that are no multiple of
    the 128 x 160 tile (clamped column vectors, partial stores), slices of one and two k-tiles -- against float64, and BIT FOR BIT
    against gemm_kernel<false,false,4,4,160>'s partial slices (txe_gemm_plain route bit 2: the same k order per element); route bit 4
    puts every shape on the 160-wide tiles."""
    from taxoexpan_amd import _lib
    dev = _dev()
    rs = np.random.RandomState(M + N)
    lda, ldb, ldc = (M + 3) // 4 * 4 + 4, (N + 3) // 4 * 4, (N + 3) // 4 * 4
    A = rs.standard_normal((K, lda)).astype(np.float32)
    B = rs.standard_normal((K, ldb)).astype(np.float32)
    At, Bt = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    outs = {}
    for tag, route in (("lds", 4), ("old", 4 | 2)):
        C = torch.full((S * M, ldc), float("nan"), device=dev)
        _lib.call("txe_gemm_plain", 2, At.data_ptr(), lda, Bt.data_ptr(), ldb, C.data_ptr(), ldc, M, N, K, S, route, None, 0, _lib.stream_ptr())
        torch.cuda.synchronize()
        outs[tag] = C.cpu().numpy().reshape(S, M, ldc)
    got = outs["lds"]
    assert np.isfinite(got[:, :, :N]).all() and (ldc == N or np.isnan(got[:, :, N:]).all())
    ref = A.astype(np.float64)[:, :M].T @ B.astype(np.float64)[:, :N]
    np.testing.assert_allclose(got[:, :, :N].astype(np.float64).sum(0), ref, rtol=1e-4, atol=1e-4 * np.sqrt(K))
    assert np.array_equal(outs["old"][:, :, :N], got[:, :, :N])


def test_forward_sweep_two_nodes_per_wave_is_bit_equal_to_one():
    """gat_aggregate_fwd_kernel with NPW = 2 (the heads of two nodes' load chains fetched together, txe_gat.hip) against NPW = 1 on a
    generic multigraph -- hubs above 64 in-edges, nodes without in-edges, an odd node count (a wave with one node), the last node a
    hub -- in all four epilogue modes (plain, the next layer's logits with and without its mask, that layer's dropout on the rows):
    out, alpha and the logits bit for bit (txe_gat_aggregate_fwd's npw argument)."""
    from taxoexpan_amd import _lib
    from taxoexpan_amd._lib import call, ptr
    rs = np.random.RandomState(3)
    N, H, D, kp = 4099, 4, 52, 224
    deg = rs.randint(0, 6, size=N); deg[[5, 77, N - 1]] = [70, 200, 130]; deg[[6, 7, 4000]] = 0
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = rs.randint(0, N, size=int(rowptr[-1])).astype(np.int32)
    dev = _dev()
    rp, cl = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    ft = torch.from_numpy(rs.standard_normal((N, H * D)).astype(np.float32)).to(dev)
    a12 = torch.from_numpy(rs.standard_normal((N, 2 * H)).astype(np.float32)).to(dev)
    wa = torch.from_numpy(rs.standard_normal((2, kp)).astype(np.float32)).to(dev)
    mask = torch.from_numpy(rs.randint(0, 2 ** 31, size=(N, kp // 32)).astype(np.int32)).to(dev)
    outs = {}
    for npw in (1, 2):
        res = {}
        for mode, (nx, nx_p, use_mask) in dict(plain=(False, 0.0, False), logits=(True, 0.0, False), logits_mask=(True, 0.5, True),
                                               rows_dropped=(False, 0.5, True)).items():
            for attn_p, keep in ((0.0, False), (0.3, True)):
                out = torch.full((N, kp), 0.25, device=dev)
                alpha = torch.full((int(rowptr[-1]) * H + 1,), -1.0, device=dev)
                nxa = torch.full((N, 2), -1.0, device=dev)
                call('txe_gat_aggregate_fwd', ptr(rp), ptr(cl), N, ptr(ft), H * D, ptr(a12), ptr(a12[:, H:]), 2 * H, H, D, 0.2, attn_p, 99, 1, 0.01,
                     ptr(out), kp, ptr(alpha) if keep else None, ptr(wa) if nx else None, kp, ptr(mask) if use_mask else None, nx_p,
                     ptr(nxa) if nx else None, npw, _lib.stream_ptr())
                torch.cuda.synchronize()
                k = mode + ('_train' if keep else '_eval')
                res[k + '_out'], res[k + '_alpha'], res[k + '_nx'] = out.cpu().numpy(), alpha.cpu().numpy(), nxa.cpu().numpy()
        outs[npw] = res
    assert len(outs[1]) == 24
    for k in outs[1]:
        assert np.isfinite(outs[1][k]).all(), k
        assert np.array_equal(outs[1][k], outs[2][k]), k
    assert not np.array_equal(outs[1]["plain_eval_out"][:, :208], np.full((4099, 208), 0.25, dtype=np.float32))


def _egonet_csr_for_walk(rs, shapes, dev):
    from taxoexpan_amd.graph import BatchedDGLGraph
    g = BatchedDGLGraph.from_egonet_shapes([s[0] for s in shapes], [s[1] for s in shapes])
    csr = g.csr(dev)
    return csr.rowptr_in, csr.col_src, csr.n_nodes, csr.n_edges


@pytest.mark.parametrize("D,kp", [(52, 224), (500, 2080), (600, 2464)])
def test_forward_sweep_walking_egonets_is_bit_equal_to_node_per_wave(D, kp):
    """gat_aggregate_ego_kernel (txe_gat.hip: a workgroup walks a window of consecutive destination nodes, every row read once) against
    gat_aggregate_fwd_kernel (one wave per node) -- `out` and `alpha` bit for bit, the next layer's logits within rounding (another
    summation order) -- on a batch of egonets (anchors without parents, with 40 parents, with 51 siblings, single nodes) for several
    window sizes (so that windows start behind anchors and between parents), and on a generic multigraph (no egonet in it: hubs above 64
    in-edges, nodes without in-edges, chains whose runs overlap), where every node takes the kernel's generic path."""
    from taxoexpan_amd import _lib
    from taxoexpan_amd._lib import call, ptr
    rs = np.random.RandomState(11)
    dev = _dev()
    H = 4
    shapes = [(int(rs.randint(0, 4)), int(rs.randint(0, 9))) for _ in range(900)]
    shapes[3], shapes[4], shapes[5], shapes[6], shapes[7] = (40, 2), (0, 0), (1, 51), (63, 0), (0, 51)
    shapes[200:210] = [(0, 0)] * 10
    graphs = {}
    rp, cl, N, E = _egonet_csr_for_walk(rs, shapes, dev)
    graphs["egonets"] = (rp, cl, N, E)
    Ng = 4099
    deg = rs.randint(0, 6, size=Ng); deg[[5, 77, Ng - 1]] = [70, 200, 130]; deg[[6, 7, 4000]] = 0
    deg[1000:1040] = 3; deg[1100:1110] = 2; deg[1200:1230] = 2; deg[1300:1320] = 2; deg[1400:1410] = 1
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = rs.randint(0, Ng, size=int(rowptr[-1])).astype(np.int32)
    for v in range(1000, 1040): col[rowptr[v]:rowptr[v + 1]] = [v - 2, v - 1, v]          # runs that overlap
    for v in range(1100, 1110): col[rowptr[v]:rowptr[v + 1]] = [v, v - 1]                 # the self loop first
    for v in range(1200, 1230): col[rowptr[v]:rowptr[v + 1]] = [v - 5, v]                 # every node another hub
    for v in range(1300, 1320): col[rowptr[v]:rowptr[v + 1]] = [1290 + (v & 1), v]        # two hubs, alternating
    for v in range(1400, 1410): col[rowptr[v]:rowptr[v + 1]] = [v + 1]                    # one in-edge, not the self loop
    graphs["multigraph"] = (torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev), Ng, int(rowptr[-1]))
    for gname, (rp, cl, N, E) in graphs.items():
        ft = torch.from_numpy(rs.standard_normal((N, H * D)).astype(np.float32)).to(dev)
        a12 = torch.from_numpy(rs.standard_normal((N, 2 * H)).astype(np.float32)).to(dev)
        wa = torch.from_numpy(rs.standard_normal((2, kp)).astype(np.float32)).to(dev)
        mask = torch.from_numpy(rs.randint(0, 2 ** 31, size=(N, kp // 32)).astype(np.int32)).to(dev)
        outs = {}
        for npw in (1, 3, 8, 13, 32):
            res = {}
            for mode, (nx, nx_p, use_mask) in dict(plain=(False, 0.0, False), logits=(True, 0.0, False), logits_mask=(True, 0.5, True),
                                                   rows_dropped=(False, 0.5, True)).items():
                for attn_p, keep in ((0.0, False), (0.3, True)):
                    out = torch.full((N, kp), 0.25, device=dev)
                    alpha = torch.full((E * H + 1,), -1.0, device=dev)
                    nxa = torch.full((N, 2), -1.0, device=dev)
                    call('txe_gat_aggregate_fwd', ptr(rp), ptr(cl), N, ptr(ft), H * D, ptr(a12), ptr(a12[:, H:]), 2 * H, H, D, 0.2, attn_p, 99, 1, 0.01,
                         ptr(out), kp, ptr(alpha) if keep else None, ptr(wa) if nx else None, kp, ptr(mask) if use_mask else None, nx_p,
                         ptr(nxa) if nx else None, npw, _lib.stream_ptr())
                    torch.cuda.synchronize()
                    k = mode + ('_train' if keep else '_eval')
                    res[k + '_out'], res[k + '_alpha'], res[k + '_nx'] = out.cpu().numpy(), alpha.cpu().numpy(), nxa.cpu().numpy()
            outs[npw] = res
        for npw in (3, 8, 13, 32):
            for k in outs[1]:
                assert np.isfinite(outs[npw][k]).all(), (gname, npw, k)
                if k.endswith('_nx'):
                    scale = np.abs(outs[1][k]).max() + 1e-6
                    assert np.abs(outs[npw][k] - outs[1][k]).max() <= 2e-6 * scale * np.sqrt(kp), (gname, npw, k)
                else:
                    assert np.array_equal(outs[1][k], outs[npw][k]), (gname, npw, k, int((outs[1][k] != outs[npw][k]).sum()))


@pytest.mark.parametrize("D,with_nx", [(500, True), (52, False), (600, True)])
def test_table_sweep_walking_egonets_is_bit_equal_to_node_per_wave(D, with_nx):
    """txe_gat_aggregate_table_fwd on the egonet walk (gat_aggregate_ego_kernel<.., TAB>: rows T[rid[u]] + T2[pos[u]] formed from the
    table inside the walk) against its wave-per-node kernel: `out` bit for bit, the next layer's logits within rounding -- a batch of
    egonets for several window sizes, and a generic multigraph (every node on the kernel's generic path)"""
    from taxoexpan_amd import _lib, graph as G
    rs = np.random.RandomState(23)
    dev = _dev()
    H, vocab, n_tab = 4, 3, 300
    F, Fe = H * D, H * D + 2 * H
    Fp = -(-Fe // 32) * 32
    kp = -(-(F + 50) // 32) * 32
    shapes = [(int(rs.randint(0, 4)), int(rs.randint(0, 9))) for _ in range(700)]
    shapes[3], shapes[4], shapes[5], shapes[6] = (40, 2), (0, 0), (1, 51), (63, 0)
    graphs = {"egonets": _egonet_csr_for_walk(rs, shapes, dev)[:3]}
    Ng = 1500
    src, dst = rs.randint(0, Ng, 5000), rs.randint(5, Ng, 5000)
    src[:150], dst[:150] = rs.randint(0, Ng, 150), 17
    rin, col = G.build_csr_device(torch.from_numpy(src.astype(np.int32)).to(dev), torch.from_numpy(dst.astype(np.int32)).to(dev), Ng)[:2]
    graphs["multigraph"] = (rin, col, Ng)
    if _lib.call("txe_gat_aggregate_table_supported", H, D, Fp, vocab, kp if with_nx else 0) != 1:
        pytest.skip("the T2 rows of this width do not fit the LDS")
    T = torch.from_numpy(rs.standard_normal((n_tab, Fp)).astype(np.float32)).to(dev)
    T2 = torch.from_numpy(rs.standard_normal((vocab, Fp)).astype(np.float32)).to(dev)
    wa = torch.from_numpy(rs.standard_normal((2, kp)).astype(np.float32)).to(dev)
    for gname, (rp, cl, N) in graphs.items():
        rid = torch.from_numpy(rs.randint(0, n_tab, N).astype(np.int32)).to(dev)
        pos = torch.from_numpy(rs.randint(0, vocab, N).astype(np.int32)).to(dev)
        ld_out = kp if with_nx else F
        outs = {}
        for npw in (1, 3, 8, 13, 32):
            out = torch.full((N, ld_out), 0.25, device=dev)
            a12 = torch.zeros(N, 2, device=dev)
            _lib.call("txe_gat_aggregate_table_fwd", rp.data_ptr(), cl.data_ptr(), N, T.data_ptr(), Fp, rid.data_ptr(), T2.data_ptr(),
                      pos.data_ptr(), vocab, H, D, 0.2, 1, 0.1, out.data_ptr(), ld_out, wa.data_ptr() if with_nx else None, kp if with_nx else 0,
                      a12.data_ptr() if with_nx else None, npw, _lib.stream_ptr())
            torch.cuda.synchronize()
            outs[npw] = (out.cpu().numpy(), a12.cpu().numpy())
        assert np.isfinite(outs[1][0]).all() and np.abs(outs[1][0][:, :F]).max() > 0
        for npw in (3, 8, 13, 32):
            assert np.array_equal(outs[1][0], outs[npw][0]), (gname, npw, int((outs[1][0] != outs[npw][0]).sum()))
            scale = np.abs(outs[1][1]).max() + 1e-6
            assert np.abs(outs[npw][1] - outs[1][1]).max() <= 2e-6 * scale * np.sqrt(kp), (gname, npw)


def test_walk_plan_of_a_batch_names_hub_roles_and_csr_positions():
    """txe_egonet_walk_plan (what the egonet-walking sweeps otherwise work out per workgroup and step) on a batch of egonets -- list order
    anchor, parents, siblings; roles; the destination-CSR positions of every node's self loop and of its edge with the anchor -- and on
    graphs that are no egonets (not walked: list position = node)"""
    from taxoexpan_amd import _lib
    from taxoexpan_amd.graph import BatchedDGLGraph, DGLGraph, batch
    rs = np.random.RandomState(5)
    dev = _dev()
    shapes = [(int(rs.randint(0, 4)), int(rs.randint(0, 9))) for _ in range(300)]
    shapes[3], shapes[4], shapes[5], shapes[6] = (40, 2), (0, 0), (1, 51), (62, 1)
    g = BatchedDGLGraph.from_egonet_shapes([s_[0] for s_ in shapes], [s_[1] for s_ in shapes])
    others = []
    for n in (5, 70, 9):                                        # a ring, a graph of more than 64 nodes, two hubs
        h = DGLGraph(); h.add_nodes(n)
        h.add_edges(np.arange(n), (np.arange(n) + 1) % n)
        if n == 9: h.add_edges(np.array([0, 0, 0, 1, 1, 1]), np.array([2, 3, 4, 5, 6, 7]))
        h.add_edges(h.nodes(), h.nodes())
        others.append(h)
    for graph, egonets in ((g, True), (batch(others), False)):
        csr = graph.csr(dev)
        N, G = csr.n_nodes, csr.n_graphs
        plan = torch.full((_lib.call("txe_egonet_walk_plan_bytes", N) // 4,), -7, dtype=torch.int32, device=dev)
        _lib.call("txe_egonet_walk_plan", csr.rowptr_in.data_ptr(), csr.col_src.data_ptr(), csr.rowptr_out.data_ptr(), csr.col_dst.data_ptr(),
                  csr.pos_out.data_ptr(), csr.graph_off.data_ptr(), N, G, plan.data_ptr(), _lib.stream_ptr())
        torch.cuda.synchronize()
        P = plan.cpu().numpy().reshape(N, 8)
        goff = csr.graph_off.cpu().numpy()
        src = csr.col_src.cpu().numpy()
        dst = np.repeat(np.arange(N), np.diff(csr.rowptr_in.cpu().numpy()))
        for gi in range(G):
            o, n = int(goff[gi]), int(goff[gi + 1] - goff[gi])
            rows = P[o:o + n]
            assert (rows[:, 5] == o).all()
            if not egonets:
                assert (rows[:, 0] == np.arange(o, o + n)).all() and ((rows[:, 1] & 31) == 0).all()      # not walked, no role
                assert ((rows[:, 1] & 32) != 0).all() == (n <= 64)
                continue
            k, m = shapes[gi]
            if (k, m) == (0, 1): k, m = 1, 0                    # (anchor -> one sibling IS parent -> anchor: the walk takes the second reading)
            anchor = o + k
            assert list(rows[:, 0]) == [anchor] + list(range(o, anchor)) + list(range(anchor + 1, o + n)), gi
            assert (rows[:, 1] >> 4 == 3).all() and (rows[:, 4] == anchor).all()
            assert list(rows[:, 1] & 15) == [1] + [2] * k + [3] * m
            for node, fl, ps, ph in rows[:, :4]:
                assert src[ps] == node and dst[ps] == node
                role = fl & 15
                if role == 1: assert ph == ps
                elif role == 2: assert src[ph] == node and dst[ph] == anchor
                else: assert src[ph] == anchor and dst[ph] == node


@pytest.mark.parametrize("N,Kh,ld_h,Pd,p,H,D", [(517, 250, 250, 50, 0.3, 4, 500), (33, 10, 10, 4, 0.3, 2, 6), (301, 251, 253, 50, 0.1, 4, 24),
                                                   (64, 96, 96, 0, 0.5, 2, 60), (129, 300, 300, 50, 0.0, 4, 600), (1, 250, 250, 50, 0.5, 4, 500)])
def test_first_layer_input_formed_inside_the_packs_equals_the_stored_one(N, Kh, ld_h, Pd, p, H, D):
    """txe_gat_dense_fwd_split_src (X = dropout([h | Emb[pos]]) never stored: the packs form it from h, the position table and the keep
    mask that txe_gat_layers_prepare writes with X == NULL) against txe_gat_layers_prepare + txe_gat_dense_fwd_split on the stored X:
    the same mask and weights, Y and the contraction-major planes (the backward pass's Xt) bit for bit -- 8-byte and 4-byte feature
    rows, a width whose last quad straddles Kh, no position columns, no dropout, a single row"""
    import ctypes
    from taxoexpan_amd import _lib
    dev = _dev()
    rs = np.random.RandomState(7 + Kh + N)
    seed = 9876543210
    Kt = Kh + Pd
    Kp, Fp = _lib.call("txe_gat_padded_k", Kh, Pd), _lib.call("txe_gat_padded_f", H, D)
    wsb = _lib.call("txe_gat_dense_split_ws_bytes", N, Kh, Pd, H, D)
    xtb = _lib.call("txe_gat_dense_split_xt_bytes", N, Kh, Pd, H, D)
    assert wsb > 0 and xtb > 0
    h = torch.from_numpy(rs.standard_normal((N, ld_h)).astype(np.float32)).to(dev)
    pos = torch.from_numpy(rs.randint(0, 3, N).astype(np.int32)).to(dev)
    P = torch.from_numpy(rs.standard_normal((3, max(Pd, 1))).astype(np.float32)).to(dev)[:, :Pd].contiguous() if Pd else None
    W = torch.from_numpy(rs.standard_normal((H * D, Kt)).astype(np.float32)).to(dev)
    al, ar = (torch.from_numpy(rs.standard_normal((1, H, D)).astype(np.float32)).to(dev) for _ in range(2))
    ptr = lambda t: t.data_ptr() if t is not None else None
    outs = []
    for stored in (True, False):
        X = torch.full((N, Kp), 7.0, device=dev)
        Wp = torch.full((Fp, Kp), 7.0, device=dev)
        mask = torch.zeros((N, (Kt + 31) // 32), dtype=torch.int32, device=dev) if p > 0 else None
        d = (_lib.GatPrepareDesc * 1)()
        d[0].h, d[0].ld_h, d[0].n_nodes, d[0].Kh, d[0].pos, d[0].P, d[0].Pd = h.data_ptr(), ld_h, N, Kh, ptr(pos) if Pd else None, ptr(P), Pd
        d[0].X, d[0].W, d[0].attn_l, d[0].attn_r, d[0].H, d[0].D, d[0].Wp = (X.data_ptr() if stored else None), W.data_ptr(), al.data_ptr(), ar.data_ptr(), H, D, Wp.data_ptr()
        d[0].feat_drop_p, d[0].seed, d[0].mask, d[0].x_dropped = p, seed, ptr(mask), 1
        _lib.call("txe_gat_layers_prepare", ctypes.cast(d, ctypes.c_void_p), 1, _lib.stream_ptr())
        Y = torch.full((N, Fp), -3.0, device=dev)
        ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
        Xt = torch.zeros(xtb, dtype=torch.uint8, device=dev)
        if stored:
            _lib.call("txe_gat_dense_fwd_split", X.data_ptr(), N, Kh, Pd, Wp.data_ptr(), H, D, None, None, Xt.data_ptr(), Y.data_ptr(), ws.data_ptr(), wsb,
                      _lib.stream_ptr())
        else:
            _lib.call("txe_gat_dense_fwd_split_src", h.data_ptr(), ld_h, ptr(pos) if Pd else None, ptr(P), ptr(mask), p, N, Kh, Pd, Wp.data_ptr(), H, D,
                      Xt.data_ptr(), Y.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr())
            assert float((X - 7.0).abs().max()) == 0.0                     # never written
        torch.cuda.synchronize()
        outs.append((Y.cpu(), Xt.cpu(), Wp.cpu(), mask.cpu() if mask is not None else None))
    (Ya, Xta, Wpa, ma), (Yb, Xtb_, Wpb, mb) = outs
    assert torch.equal(Wpa, Wpb) and (ma is None or torch.equal(ma, mb))
    assert torch.equal(Ya[:, :H * D + 2 * H], Yb[:, :H * D + 2 * H]) and bool(torch.isfinite(Ya[:, :H * D + 2 * H]).all())
    assert torch.equal(Xta, Xtb_)


def test_readout_and_match_ops_against_oracle():
    from taxoexpan_amd import ops
    from taxoexpan_amd.graph import BatchedDGLGraph
    rs = np.random.RandomState(5)
    shapes = [(0, 0), (2, 3), (1, 50), (3, 0), (0, 7)] * 5
    g = BatchedDGLGraph.from_egonet_shapes([s[0] for s in shapes], [s[1] for s in shapes])
    graph = orc.batch_egonets(shapes)
    N = g.number_of_nodes()
    for D in (6, 500, 33):
        h = torch.from_numpy(rs.standard_normal((N, D)).astype(np.float32))
        pw = torch.from_numpy(rs.standard_normal((3, 1)).astype(np.float32))
        w = torch.from_numpy(rs.standard_normal((len(shapes), D)).astype(np.float32))
        for weighted in (True, False):
            hg_d = h.to(_dev()).requires_grad_(True)
            pw_d = pw.to(_dev()).requires_grad_(True)
            out = ops.ReadoutFunction.apply(g.csr(_dev()), hg_d, g.ndata["pos"].to(_dev()), pw_d if weighted else None)
            (out * w.to(_dev())).sum().backward()
            hc, pc = h.clone().requires_grad_(True), pw.clone().requires_grad_(True)
            ref = orc.weighted_mean_readout(graph["graph_off"], hc, graph["pos"], pc) if weighted else orc.mean_readout(graph["graph_off"], hc)
            (ref * w).sum().backward()
            np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=RT, atol=AT)
            np.testing.assert_allclose(hg_d.grad.cpu().numpy(), hc.grad.numpy(), rtol=1e-4, atol=1e-6)
            if weighted:
                np.testing.assert_allclose(pw_d.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-3, atol=1e-5)
    # pairwise bilinear, both BIM and LBM, with gradient to both sides
    G, l, r = 37, 50, 23
    e1 = torch.from_numpy(rs.standard_normal((G, l)).astype(np.float32) * 0.3)
    e2 = torch.from_numpy(rs.standard_normal((G, r)).astype(np.float32) * 0.3)
    W = torch.from_numpy(rs.standard_normal((1, l, r)).astype(np.float32) * 0.2)
    up = torch.from_numpy(rs.standard_normal((G, 1)).astype(np.float32))
    for ex in (False, True):
        for query_grad in (True, False):          # False: the query-side form (V = e2 W^T, elementwise d_e1) that training takes
            a, b, c = (t.to(_dev()).requires_grad_(True) for t in (e1, e2, W))
            b.requires_grad_(query_grad)
            s = ops.BilinearPairFunction.apply(a, b, c, ex)
            (s * up.to(_dev())).sum().backward()
            ac, bc, cc = (t.clone().requires_grad_(True) for t in (e1, e2, W))
            sr = orc.bilinear_match(ac, bc, cc, ex)
            (sr * up).sum().backward()
            np.testing.assert_allclose(s.detach().cpu().numpy(), sr.detach().numpy(), rtol=RT, atol=AT)
            for got, want in ((a, ac), (b, bc), (c, cc)) if query_grad else ((a, ac), (c, cc)):
                np.testing.assert_allclose(got.grad.cpu().numpy(), want.grad.numpy(), rtol=1e-3, atol=1e-5)
            assert query_grad or b.grad is None


def test_query_projection_prefetched_on_the_second_stream_changes_nothing(monkeypatch):
    """ops.bilinear_query_prefetch: the matcher's query projection runs on the second stream under the encoder (started behind the first
    projection GEMM) or in line (no second stream): same scores and gradients, bit for bit"""
    from taxoexpan_amd import ops
    name = next(n for n in NODROP if gc.CASES[n]["match"] in ("LBM", "BIM"))
    spec, z, shapes, x, q, params, graph = load_case(name)
    model = _build_model(spec, params).eval()
    outs = []
    monkeypatch.setattr(ops, "_NO_QUERY_RUNS", True)                # (the GEMM form of the match, whatever the batch size)
    for no_side in (True, False):
        monkeypatch.setattr(ops, "_NO_SIDE_STREAM", no_side)
        model.zero_grad(set_to_none=True)
        s = model(_graph(shapes), torch.from_numpy(x).to(_dev()), torch.from_numpy(q).to(_dev()))
        s.sum().backward()
        torch.cuda.synchronize()
        outs.append((s.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0])
        assert outs[0][1].keys() == o[1].keys() and all(torch.equal(outs[0][1][k], o[1][k]) for k in o[1])


def test_scoring_loop_and_ranks_against_reference_goldens():
    from taxoexpan_amd import ops
    from taxoexpan_amd.model_zoo import BIM, LBM
    from taxoexpan_amd.scoring import score_all
    z = dict(np.load(f"{GOLDEN_DIR}/scoring.npz"))
    hg, qs, W, positives = gc.make_scoring_inputs()
    pos_off = np.cumsum([0] + [len(p) for p in positives]).astype(np.int32)
    pos_idx = np.concatenate(positives).astype(np.int32)
    for kind, cls in (("lbm", LBM), ("bim", BIM)):
        mod = cls(hg.shape[1], qs.shape[1])
        mod.load_state_dict({"W.weight": torch.from_numpy(W)})
        mod = mod.to(_dev())
        S = score_all(mod, torch.from_numpy(hg).to(_dev()), torch.from_numpy(qs).to(_dev()), block=5)
        np.testing.assert_allclose(S.cpu().numpy(), z[f"S_{kind}"], rtol=RT, atol=AT)
        # literal per-query call pattern of test_fast.py:121-123 through the module's forward
        with torch.no_grad():
            nf = torch.from_numpy(qs[3]).to(_dev())
            e = mod(torch.from_numpy(hg).to(_dev()), nf.expand(hg.shape[0], -1))
        np.testing.assert_allclose(e.squeeze(1).cpu().numpy(), z[f"S_{kind}"][3], rtol=RT, atol=AT)
        # rank kernel on the REFERENCE's scores: integer-exact against metric.py's ranks
        ranks = ops.rank_block(torch.from_numpy(z[f"S_{kind}"]).to(_dev()), torch.from_numpy(pos_off), torch.from_numpy(pos_idx), True)
        assert ranks.cpu().numpy().tolist() == z[f"ranks_{kind}"].tolist()
        # and end to end on our own scores: ranks may only differ where scores tie within fp32 noise -- require equality
        ranks2 = ops.rank_block(S, torch.from_numpy(pos_off), torch.from_numpy(pos_idx), True)
        assert ranks2.cpu().numpy().tolist() == z[f"ranks_{kind}"].tolist()


def test_device_csr_build_equals_host_build():
    from taxoexpan_amd.graph import DGLGraph
    src, dst = _random_graph(1000, 20000, seed=9)
    g = DGLGraph()
    g.add_nodes(1000)
    g.add_edges(src, dst)
    a = g.csr("cpu", method="host")
    g._csr_cache.clear()
    b = g.csr(_dev(), method="device")
    for f in ("rowptr_in", "col_src", "eid_in", "rowptr_out", "col_dst", "pos_out"):
        assert torch.equal(getattr(a, f), getattr(b, f).cpu()), f
    # empty graph
    g0 = DGLGraph()
    g0.add_nodes(5)
    c = g0.csr(_dev(), method="device")
    assert c.rowptr_in.cpu().tolist() == [0] * 6


def test_full_size_properties_mag_cs_batch():
    """BASELINE.json configs[1] size (4,096 egonets, MAG dims): size-independent properties instead of an oracle run:
    attention rows sum to one per destination, bitwise repeatability (atomic-free reductions), linearity of the
    aggregation in ft.  (The oracle comparison of the whole step at this size lives in tests/test_gpu_full_size.py.)"""
    from taxoexpan_amd import _lib, synthetic as syn
    from taxoexpan_amd.ops import _empty
    tax = syn.make_named_taxonomy("mag_cs")
    g, qf, labels = syn.training_batch(tax, 128, 31, seed=3)
    dev = _dev()
    csr = g.csr(dev)
    N, E, H, D = csr.n_nodes, csr.n_edges, 4, 500
    torch.manual_seed(0)
    ft = torch.randn(N, H * D, device=dev)
    a = torch.randn(N, 2 * H, device=dev)

    def agg(ft_):
        out, alpha = _empty((N, H * D), ft_), _empty((E, H), ft_)
        _lib.call("txe_gat_aggregate_fwd", csr.rowptr_in.data_ptr(), csr.col_src.data_ptr(), N, ft_.data_ptr(), H * D, a.data_ptr(),
                  a.data_ptr() + 4 * H, 2 * H, H, D, 0.2, 0.0, 0, 0, 1.0, out.data_ptr(), H * D, alpha.data_ptr(), None, 0, None, 0.0, None,
                  0, _lib.stream_ptr())
        return out, alpha
    out1, alpha1 = agg(ft)
    out2, alpha2 = agg(ft)
    assert torch.equal(out1, out2) and torch.equal(alpha1, alpha2)                 # bitwise repeatable
    seg = torch.zeros(N, H, device=dev).index_add_(0, torch.repeat_interleave(torch.arange(N, device=dev), (csr.rowptr_in[1:] - csr.rowptr_in[:-1]).long()), alpha1)
    np.testing.assert_allclose(seg.cpu().numpy(), 1.0, rtol=1e-5)                 # softmax over in-edges
    ft_b = torch.randn(N, H * D, device=dev)
    out_b, _ = agg(ft_b)
    out_ab, _ = agg(2.0 * ft + ft_b)
    np.testing.assert_allclose(out_ab.cpu().numpy(), (2.0 * out1 + out_b).cpu().numpy(), rtol=1e-4, atol=1e-4)   # linear in ft
    assert E == 2 * N - g.batch_size                                               # E = 2n-1 per egonet


def test_sharded_scoring_single_rank_rccl():
    """the candidate-sharded scoring path with its real (HIP) local scorer and an RCCL all-gather, world size 1 (the only
    size a 1-GPU box offers; world size 2 is covered on CPU/gloo in tests/test_distributed_cpu.py)"""
    import os
    import torch.distributed as dist
    from taxoexpan_amd.model_zoo import LBM
    from taxoexpan_amd.scoring import allreduce_gradients, score_all, score_all_sharded, shard_bounds
    z = dict(np.load(f"{GOLDEN_DIR}/scoring.npz"))
    hg, qs, W, positives = gc.make_scoring_inputs()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=_dev())
        created = True
    try:
        mod = LBM(hg.shape[1], qs.shape[1])
        mod.load_state_dict({"W.weight": torch.from_numpy(W)})
        mod = mod.to(_dev())
        lo, hi = shard_bounds(hg.shape[0], 1, 0)
        S = score_all_sharded(mod, torch.from_numpy(hg[lo:hi]).to(_dev()), hg.shape[0], torch.from_numpy(qs).to(_dev()), block=7)
        np.testing.assert_allclose(S.cpu().numpy(), z["S_lbm"], rtol=RT, atol=AT)
        assert torch.equal(S, score_all(mod, torch.from_numpy(hg).to(_dev()), torch.from_numpy(qs).to(_dev()), block=7))
        p = torch.nn.Parameter(torch.ones(5, device=_dev()))
        p.grad = torch.arange(5.0, device=_dev())
        allreduce_gradients([p])
        assert torch.equal(p.grad.cpu(), torch.arange(5.0))
    finally:
        if created:
            dist.destroy_process_group()


def test_sum_max_readouts_against_oracle():
    from taxoexpan_amd.graph import BatchedDGLGraph
    from taxoexpan_amd.model_zoo import MaxReadout, SumReadout
    rs = np.random.RandomState(11)
    shapes = [(0, 0), (2, 3), (1, 50), (3, 0), (0, 7)] * 3
    g = BatchedDGLGraph.from_egonet_shapes([s[0] for s in shapes], [s[1] for s in shapes])
    graph = orc.batch_egonets(shapes)
    N, D = g.number_of_nodes(), 37
    h = torch.from_numpy(rs.standard_normal((N, D)).astype(np.float32))
    w = torch.from_numpy(rs.standard_normal((len(shapes), D)).astype(np.float32))
    for mod, ref_fn in ((SumReadout(), orc.sum_readout), (MaxReadout(), orc.max_readout)):
        hd = h.to(_dev()).requires_grad_(True)
        g.ndata["h"] = hd
        out = mod(g)
        (out * w.to(_dev())).sum().backward()
        hc = h.clone().requires_grad_(True)
        ref = ref_fn(graph["graph_off"], hc)
        (ref * w).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=RT, atol=AT)
        np.testing.assert_allclose(hd.grad.cpu().numpy(), hc.grad.numpy(), rtol=1e-4, atol=1e-6)


def test_device_egonet_builder_equals_host_builder():
    """txe_egonet_* (dataset.py:404-437 + dgl.batch on device) against the host builder: bit-exact node table and both CSR
    views when no anchor exceeds expand_factor; with sampling, every property the reference's construction guarantees."""
    from taxoexpan_amd import graph as G
    from taxoexpan_amd import synthetic as syn
    dev = _dev()
    tax = syn.make_taxonomy(6000, 9500, 12, seed=5)
    dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
    rs = np.random.RandomState(1)
    anchors = rs.randint(0, tax.n_nodes, 777)
    anchors[:3] = [0, tax.n_nodes - 1, int(np.argmax(np.diff(tax.chd_ptr)))]       # root (k=0), a leaf (m=0), the widest node
    exclude = np.full(anchors.size, -1, dtype=np.int64)
    for i, a in enumerate(anchors):                                                # every third egonet drops one of its children
        ch = tax.chd_idx[tax.chd_ptr[a]:tax.chd_ptr[a + 1]]
        if i % 3 == 0 and ch.size:
            exclude[i] = ch[rs.randint(ch.size)]
    big = 10 ** 6
    for ex in (None, exclude):
        hg = syn.egonet_batch(tax, anchors, expand_factor=big, exclude_child=ex)
        dg = G.device_egonet_batch(dtax, anchors, exclude=ex, expand_factor=big)
        hc, dc = hg.csr("cpu", "host"), dg.csr(dev)
        assert (dc.n_nodes, dc.n_edges, dc.n_graphs) == (hc.n_nodes, hc.n_edges, hc.n_graphs)
        assert torch.equal(dg.ndata["_id"].cpu().long(), hg.ndata["_id"]) and torch.equal(dg.ndata["pos"].cpu().long(), hg.ndata["pos"].long())
        for f in ("rowptr_in", "col_src", "eid_in", "rowptr_out", "col_dst", "pos_out", "graph_off"):
            assert torch.equal(getattr(dc, f).cpu(), getattr(hc, f)), f
        assert torch.equal(dg.ndata["x"].cpu(), hg.ndata["x"])
        assert dg.batch_num_nodes == hg.batch_num_nodes and np.array_equal(dg._src, hg._src) and np.array_equal(dg._dst, hg._dst)
    # sampling with replacement (dataset.py:419): exactly expand_factor draws from the anchor's children, minus the excluded
    dg = G.device_egonet_batch(dtax, anchors, exclude=exclude, expand_factor=4, seed=11)
    dg2 = G.device_egonet_batch(dtax, anchors, exclude=exclude, expand_factor=4, seed=11)
    assert torch.equal(dg.ndata["_id"], dg2.ndata["_id"])                           # counter-based: reproducible
    ids, pos, off = dg.ndata["_id"].cpu().numpy(), dg.ndata["pos"].cpu().numpy(), dg.csr(dev).graph_off.cpu().numpy()
    for i, a in enumerate(anchors):
        p, d = pos[off[i]:off[i + 1]], ids[off[i]:off[i + 1]]
        k, m = int((p == 0).sum()), int((p == 2).sum())
        ch = tax.chd_idx[tax.chd_ptr[a]:tax.chd_ptr[a + 1]]
        assert d[k] == a and p[k] == 1 and np.array_equal(d[:k], tax.par_idx[tax.par_ptr[a]:tax.par_ptr[a + 1]])
        assert set(d[k + 1:].tolist()) <= set(ch.tolist()) and exclude[i] not in d[k + 1:]
        if ch.size <= 4:
            assert m == ch.size - int(exclude[i] >= 0)
        else:
            assert m <= 4 and (exclude[i] >= 0 or m == 4)
    # the device-built batch drives the encoder to the same result as the host-built one
    spec = gc.CASES["mag_pgat_wmr_lbm"] if "mag_pgat_wmr_lbm" in gc.CASES else gc.CASES[NODROP[0]]
    tax2 = syn.make_taxonomy(3000, 4700, spec["in_dim"], seed=2)
    dtax2 = G.DeviceTaxonomy(tax2.par_ptr, tax2.par_idx, tax2.chd_ptr, tax2.chd_idx, tax2.features, dev)
    model = _build_model(spec, gc.make_params(spec)).eval()
    an = np.arange(0, 3000, 7)
    hg, dg = syn.egonet_batch(tax2, an, expand_factor=big), G.device_egonet_batch(dtax2, an, expand_factor=big)
    from taxoexpan_amd.scoring import encode_candidates
    assert torch.equal(encode_candidates(model, hg), encode_candidates(model, dg))


def test_device_egonets_from_raw_dataset_files():
    """raw .terms/.taxo/.embed -> MaskedGraphDataset -> device egonet builder == the dataset's own `_get_subgraph(-1, a, 0)`
    (the all-candidate construction of test_fast.py:93-97), and the encoder gives identical candidate embeddings"""
    import os
    import shutil
    import tempfile
    from taxoexpan_amd import graph as G
    from taxoexpan_amd.dataset import MAGDataset, MaskedGraphDataset
    from taxoexpan_amd.graph import batch
    from taxoexpan_amd.scoring import encode_candidates
    dev = _dev()
    d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", None))
    try:
        for fn in os.listdir(os.path.join(GOLDEN_DIR, "toy_taxo")):
            shutil.copy(os.path.join(GOLDEN_DIR, "toy_taxo", fn), d)
        ds = MaskedGraphDataset(MAGDataset("toy", d, raw=True), mode="test", sampling_mode=0, expand_factor=50, normalize_embed=True)
    finally:
        shutil.rmtree(d)
    anchors = list(ds.graph.nodes)
    host = batch([ds._get_subgraph(-1, a, 0) for a in anchors])
    dg = G.device_egonet_batch(ds.device_taxonomy(dev), anchors, expand_factor=50)
    assert torch.equal(dg.ndata["_id"].cpu().long(), host.ndata["_id"]) and torch.equal(dg.ndata["pos"].cpu().long(), host.ndata["pos"])
    assert np.array_equal(dg._src, host._src) and np.array_equal(dg._dst, host._dst)
    torch.manual_seed(0)
    from taxoexpan_amd import TaxoExpan
    model = TaxoExpan("PGAT", "WMR", "LBM", in_dim=8, hidden_dim=16, out_dim=12, pos_dim=4, num_layers=1, heads=[2, 1], feat_drop=0.1,
                      attn_drop=0.1, hidden_drop=0.1, out_drop=0.1).to(dev).eval()
    assert torch.equal(encode_candidates(model, host), encode_candidates(model, dg))


def test_ntn_and_residual_gat_against_reference_goldens():
    """NTN matcher (model_zoo.py:331-346) and GATLayer(residual=True) (:98-103) through the HIP ops vs the reference goldens"""
    import os
    from taxoexpan_amd import model_zoo as mz
    dev = _dev()
    z = np.load(os.path.join(GOLDEN_DIR, "extras.npz"))
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    ntn = mz.NTN(12, 7, k=4)
    ntn.load_state_dict({k: torch.from_numpy(z["ntn.p." + k]) for k in ("u_R.weight", "W.weight", "W.bias", "V.weight")}, strict=True)
    ntn = ntn.to(dev)
    e1, e2 = t("ntn.e1").requires_grad_(), t("ntn.e2").requires_grad_()
    out = ntn(e1, e2)
    (out * t("ntn.coef")).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["ntn.out"], rtol=RT, atol=AT)
    np.testing.assert_allclose(e1.grad.cpu().numpy(), z["ntn.d_e1"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(e2.grad.cpu().numpy(), z["ntn.d_e2"], rtol=2e-3, atol=2e-5)
    for k, p in ntn.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), z["ntn.g." + k], rtol=2e-3, atol=2e-5, err_msg=k)
    for tag, (din, dout, H) in {"res_fc": (10, 6, 3), "res_id": (6, 6, 2)}.items():
        layer = mz.GATLayer(din, dout, H, feat_drop=0.0, attn_drop=0.0, residual=True)
        keys = ["fc.weight", "attn_l", "attn_r"] + (["res_fc.weight"] if tag == "res_fc" else [])
        layer.load_state_dict({k: torch.from_numpy(z[f"{tag}.p.{k}"]) for k in keys}, strict=True)
        layer = layer.to(dev)
        x = t(tag + ".x").requires_grad_()
        out = layer(_graph(gc.EDGE_SHAPES), x)
        (out * t(tag + ".coef")).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), z[tag + ".out"], rtol=RT, atol=AT)
        np.testing.assert_allclose(x.grad.cpu().numpy(), z[tag + ".d_x"], rtol=2e-3, atol=2e-5)
        for k, p in layer.named_parameters():
            np.testing.assert_allclose(p.grad.cpu().numpy(), z[f"{tag}.g.{k}"], rtol=2e-3, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("G,pitch,larger", [(1000, 1000, True), (1003, 1003, True), (1003, 1004, False), (5000, 5000, False), (37, 40, True)])
def test_rank_kernel_against_metric_definition(G, pitch, larger):
    """txe_rank_block on both row layouts (16-byte pitch -> single-sweep path, odd pitch -> scalar path), 1..9 positives per
    query, heavy ties: rank(p) = 1 + #{g not a positive : S[g] strictly better than S[p]}  (model/metric.py:7-31)"""
    from taxoexpan_amd import ops
    rs = np.random.RandomState(G + pitch)
    nq = 23
    S = np.round(rs.randn(nq, G) * 3).astype(np.float32) / 2            # many exact ties
    npos = rs.randint(1, 10, size=nq)
    pos_idx = np.concatenate([rs.choice(G, size=k, replace=False) for k in npos]).astype(np.int32)
    pos_off = np.concatenate([[0], np.cumsum(npos)]).astype(np.int32)
    want = []
    for q in range(nq):
        P = pos_idx[pos_off[q]:pos_off[q + 1]]
        neg = np.ones(G, dtype=bool)
        neg[P] = False
        for p in P:
            want.append(1 + int(((S[q][neg] > S[q][p]) if larger else (S[q][neg] < S[q][p])).sum()))
    Sd = torch.empty((nq, pitch), device=_dev())[:, :G]
    Sd.copy_(torch.from_numpy(S))
    got = ops.rank_block(Sd, torch.from_numpy(pos_off), torch.from_numpy(pos_idx), larger)
    assert got.cpu().numpy().tolist() == want


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the driver's keys plus `roofline` and `cpu_baseline` objects (short run)"""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extra"], cwd=repo,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    assert len(lines[0]) < 4096, len(lines[0])          # the driver keeps an 8 KB stdout tail: the whole line must fit well inside it
    compact = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "matrix_pipe", "value_fp32_mfma"):
        assert k in compact, k
    assert all(not isinstance(v, (dict, list)) for k in ("roofline", "cpu_baseline") for v in compact[k].values())      # flat objects
    assert "workload" in compact["config"] and "routes" in compact["config"] and compact["config"]["parallelism"] == "dp1"
    assert compact["matrix_pipe"].startswith("bf16x3-split") and 0.5 * compact["value"] < compact["value_fp32_mfma"] < 1.2 * compact["value"]
    # everything else (roofline_all, extras, A/B legs, prose) is in bench_extra.json and on stderr -- the same run's full record
    full_lines = [ln for ln in out.stderr.splitlines() if ln.startswith("BENCH_FULL ")]
    assert len(full_lines) == 1
    d = json.loads(full_lines[0][len("BENCH_FULL "):])
    with open(os.path.join(repo, "bench_extra.json")) as f:
        assert json.load(f) == d
    for k in ("value", "ms_per_step", "steps", "warmup", "n_gpus", "dtype"):
        assert compact[k] == d[k], k
    assert abs(compact["roofline"]["frac"] - d["roofline"]["frac"]) < 1e-5 and compact["roofline"]["kernel"] == d["roofline"]["kernel"]
    assert compact["cpu_baseline"]["cores"] == d["cpu_baseline"]["cores"] and "sample" in compact["cpu_baseline"]
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["avg_edges_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    for k in ("hbm_kernel", "hbm_frac", "hbm_avg_us", "hbm_frac_of_copy_ceiling", "copy_ceiling_gbs", "hbm_aggregate_fwd_frac",
              "mfma_main_stream_frac"):                           # flat scalars: the HBM story survives consumers that drop nested objects
        assert isinstance(r[k], (int, float, str)), k
    if os.environ.get("TXE_TEST_ROUTE", "") not in ("no_fold", "no_fused_bwd"):    # (kernels of the default route)
        assert 0.0 < r["hbm_fused_bwd_frac"] < 1.0 and 0.0 < r["hbm_dx_pos_frac"] < 1.0
    assert 0.0 < r["hbm_frac"] < 1.0 and 0.0 < r["mfma_main_stream_frac"] < 1.0
    # a fresh device-built batch inside every step (trainer.py:44-61's real per-step cost) is reported next to the resident-input value
    assert d["step_incl_batch_build_ms"] >= d["ms_per_step"] * 0.9 and d["batch_build_ms"] > 0.0
    assert 0.5 * d["ms_per_step"] < d["step_repeated_queries_ms"] < 1.5 * d["ms_per_step"] and d["step_incl_batch_build_repeated_queries_ms"] > 0.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and "sample" in c


@pytest.mark.parametrize("kind,readout,drop", [("PGAT", "WMR", 0.0), ("PGAT", "MR", 0.0), ("PGAT", "WMR", 0.2), ("PGCN", "MR", 0.0),
                                               ("PGCN", "WMR", 0.0), ("PGCN", "WMR", 0.2)])
def test_collapsed_output_layer_equals_unfused_path(kind, readout, drop):
    """PGAT / PGCN + (weighted) mean readout: the folded output layer (txe_gat_collapse_* / txe_gcn_collapse_*, G graph rows) against the ordinary
    projection -> aggregation -> readout path (N node rows) of the SAME modules, same dropout seeds -- on generic batched graphs
    (degree > 64 hub, graphs of 1..40 nodes), forward and every gradient."""
    from taxoexpan_amd import model_zoo as mz, ops
    from taxoexpan_amd.graph import DGLGraph, batch
    dev = _dev()
    rs = np.random.RandomState(3)
    graphs = []
    for i, n in enumerate([1, 2, 40, 7, 3, 90, 5]):
        g = DGLGraph()
        g.add_nodes(n)
        if n > 1:
            e = 3 * n
            g.add_edges(rs.randint(0, n, e), rs.randint(0, n, e))
        if n == 90:
            g.add_edges(rs.randint(0, n, 100), np.full(100, 11))        # hub: in-degree > 64
        g.add_edges(g.nodes(), g.nodes())
        graphs.append(g)
    bg = batch(graphs)
    N = bg.number_of_nodes()
    pos = torch.from_numpy(rs.randint(0, 3, N)).to(dev)
    x = torch.randn(N, 10, generator=torch.Generator().manual_seed(0)).to(dev)
    coef = torch.randn(len(graphs), 6, generator=torch.Generator().manual_seed(1)).to(dev)
    torch.manual_seed(5)
    if kind == "PGAT":
        prop = mz.PGAT(10, 8, 6, 4, num_layers=1, heads=[3, 1], activation=torch.nn.functional.leaky_relu, feat_drop=drop, attn_drop=drop)
    else:
        prop = mz.PGCN(10, 8, 6, 4, num_layers=1, activation=torch.nn.functional.leaky_relu, in_dropout=drop, hidden_dropout=drop,
                       output_dropout=drop)
    prop = prop.to(dev)
    ro = (mz.WeightedMeanReadout() if readout == "WMR" else mz.MeanReadout()).to(dev)
    prop.train(drop > 0)
    xg = x.clone().requires_grad_(True)
    bg.ndata["pos"] = pos
    h = prop(bg, xg)                                                     # one forward: both paths share its dropout seeds
    assert isinstance(h, mz.DeferredNodeOutput)
    results = []
    for fused in (True, False):
        for p in list(prop.parameters()) + list(ro.parameters()):
            p.grad = None
        xg.grad = None
        bg.ndata["h"] = h if fused else h.tensor()
        hg = ro(bg, pos)
        (hg * coef).sum().backward()
        results.append((hg.detach().cpu().numpy(), xg.grad.cpu().numpy(),
                        {k: p.grad.cpu().numpy() for k, p in list(prop.named_parameters()) + list(ro.named_parameters())}))
    (a, dxa, ga), (b, dxb, gb) = results
    np.testing.assert_allclose(a, b, rtol=RT, atol=AT)
    np.testing.assert_allclose(dxa, dxb, rtol=2e-3, atol=2e-5)
    for k in ga:
        np.testing.assert_allclose(ga[k], gb[k], rtol=2e-3, atol=2e-5, err_msg=k)


def test_deferred_node_output_behaves_like_the_tensor():
    """graph_propagate's deferred result: arithmetic, torch functions, indexing and .cpu() all see the ordinary N x D tensor"""
    from taxoexpan_amd import model_zoo as mz
    spec, z, shapes, x, q, params, graph = load_case("small_pgat_wmr_lbm")
    model = _build_model(spec, params).eval()
    g = _graph(shapes)
    with torch.no_grad():
        h = model.graph_propagate(g, torch.from_numpy(x).to(_dev()))
        assert isinstance(h, mz.DeferredNodeOutput)
        t = h.tensor()
        assert tuple(h.shape) == tuple(t.shape) and h.device == t.device and len(h) == t.shape[0]
        assert torch.equal(h + 1.0, t + 1.0) and torch.equal(2.0 * h, 2.0 * t) and torch.equal(torch.relu(h), torch.relu(t))
        assert torch.equal(h[3], t[3]) and torch.equal(h.cpu(), t.cpu()) and torch.equal(torch.cat((h, h), 0), torch.cat((t, t), 0))


@pytest.mark.parametrize("G,r,larger,exp", [(5000, 250, True, True), (1003, 64, False, False), (37, 10, True, False)])
def test_fused_score_and_rank_equals_materialised_path(G, r, larger, exp):
    """txe_score_count_block + txe_rank_finalize (no score matrix) == txe_score_block + txe_rank_block, integer-exact, incl. ties"""
    from taxoexpan_amd import ops, scoring
    dev = _dev()
    gen = torch.Generator().manual_seed(G)
    nq = 300
    U = (torch.randn(G, r, generator=gen) * (0.05 if exp else 1.0)).to(dev)
    U[G // 2] = U[G // 3]                                    # exact score ties between two candidates
    Q = torch.randn(nq, r, generator=gen).to(dev)
    rs = np.random.RandomState(1)
    npos = rs.randint(1, 6, size=nq)
    pos_idx = np.concatenate([rs.choice(G, size=k, replace=False) for k in npos]).astype(np.int64)
    pos_idx[0] = G // 2
    pos_off = np.concatenate([[0], np.cumsum(npos)]).astype(np.int64)
    S = ops.score_block(Q, U, exp)
    want = ops.rank_block(S, torch.from_numpy(pos_off), torch.from_numpy(pos_idx), larger).cpu().tolist()
    thr = ops.positive_scores(Q, U, exp, torch.from_numpy(pos_off), torch.from_numpy(pos_idx))
    assert torch.equal(thr, S[torch.repeat_interleave(torch.arange(nq), torch.from_numpy(npos)).to(dev), torch.from_numpy(pos_idx).to(dev)])
    counts = ops.score_count_block(Q, U, exp, torch.from_numpy(pos_off), thr, larger)
    got = ops.rank_finalize(torch.from_numpy(pos_off), thr, counts, larger).cpu().tolist()
    assert got == want
    # the thresholds through the staircase of tiles only (txe_score_positives): the same bits as the full block's entries
    thr2 = torch.full((len(pos_idx) + 3,), float("nan"), device=dev)
    off32 = torch.from_numpy(pos_off.astype(np.int32)).to(dev)
    ops.positive_scores_staircase(Q, U.index_select(0, torch.from_numpy(pos_idx).to(dev)), exp, off32, thr2)
    assert torch.equal(thr2[:len(pos_idx)], thr) and bool(torch.isnan(thr2[len(pos_idx):]).all())

    class M:                                                  # the loop over query blocks, with a bilinear matcher
        apply_exp = exp
        W = type("W", (), {})()
    M.W.weight = torch.eye(r, device=dev).reshape(1, r, r)   # identity bilinear: U = hg
    got2 = scoring.rank_all_fused(M, U, Q, pos_off, pos_idx, block=128, larger_is_better=larger).cpu().tolist()
    assert got2 == want
    assert scoring.rank_all_fused(M, U, Q, pos_off, pos_idx, larger_is_better=larger).cpu().tolist() == want      # one block


@pytest.mark.parametrize("prop", ["PGAT", "PGCN"])
def test_empty_and_single_node_batches(prop):
    """edge cases of the batched input: no egonet at all, and egonets that are a lone anchor (k = m = 0: one self loop)"""
    from taxoexpan_amd import TaxoExpan
    from taxoexpan_amd.graph import BatchedDGLGraph
    dev = _dev()
    torch.manual_seed(3)
    model = TaxoExpan(prop, "WMR", "LBM", in_dim=6, hidden_dim=8, out_dim=5, pos_dim=3, num_layers=1, heads=[2, 1], feat_drop=0.0,
                      attn_drop=0.0, hidden_drop=0.0, out_drop=0.0).to(dev)
    # empty batch: shapes flow through, gradients are zeros
    g0 = BatchedDGLGraph.from_egonet_shapes([], [])
    s0 = model(g0, torch.zeros(0, 6, device=dev), torch.zeros(0, 6, device=dev))
    assert tuple(s0.shape) == (0, 1)
    if s0.requires_grad:
        s0.sum().backward()
    assert all(p.grad is None or torch.all(p.grad == 0) for p in model.parameters())
    # lone anchors: softmax over one edge is 1, the readout of one node is the node
    model.zero_grad()
    g1 = BatchedDGLGraph.from_egonet_shapes([0, 0, 0], [0, 0, 0])
    x = torch.randn(3, 6, device=dev)
    q = torch.randn(3, 6, device=dev)
    s1 = model(g1, x, q)
    assert tuple(s1.shape) == (3, 1) and torch.isfinite(s1).all()
    s1.sum().backward()
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    graph = orc.batch_egonets([(0, 0)] * 3)
    ref, _, _ = orc.taxoexpan_forward(P, graph, x.cpu(), q.cpu(), prop, "WMR", "LBM", [2, 1], 1, None)
    ref.sum().backward()
    np.testing.assert_allclose(s1.detach().cpu().numpy(), ref.detach().numpy(), rtol=RT, atol=AT)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), P[k].grad.numpy(), rtol=2e-3, atol=2e-5, err_msg=k)


def test_end_to_end_evaluation_on_raw_files_against_oracle():
    """raw .terms/.taxo/.embed -> MaskedGraphDataset(test) -> device egonets -> encode -> fused ranking -> test_fast.py metrics,
    against the same flow done literally on the host with the oracle (test_fast.py:93-133 + metric.py)"""
    import os
    import shutil
    import tempfile
    from taxoexpan_amd import TaxoExpan
    from taxoexpan_amd.dataset import MAGDataset, MaskedGraphDataset
    from taxoexpan_amd.evaluate import evaluate
    d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", None))
    try:
        for fn in os.listdir(os.path.join(GOLDEN_DIR, "toy_taxo")):
            shutil.copy(os.path.join(GOLDEN_DIR, "toy_taxo", fn), d)
        ds = MaskedGraphDataset(MAGDataset("toy", d, raw=True), mode="test", sampling_mode=0, expand_factor=100, normalize_embed=True)
    finally:
        shutil.rmtree(d)
    torch.manual_seed(11)
    heads = [2, 1]
    model = TaxoExpan("PGAT", "WMR", "LBM", in_dim=8, hidden_dim=6, out_dim=5, pos_dim=3, num_layers=1, heads=heads, feat_drop=0.1,
                      attn_drop=0.1, hidden_drop=0.1, out_drop=0.1).to(_dev())
    metrics, ranks, pos_off, queries = evaluate(model, ds, _dev())
    # host: the reference's loop with the oracle model
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cand = sorted(ds.all_positions)
    shapes, ids = [], []
    for a in cand:
        nodes, k = ds._build_egonet(-1, a, 0)
        shapes.append((k, len(nodes) - k - 1))
        ids += nodes
    graph = orc.batch_egonets(shapes)
    x = ds.node_features[torch.tensor(ids)]
    hn = orc.pgat_forward(P, graph, x, heads, 1, prefix="graph_propagate.")
    hg = orc.weighted_mean_readout(graph["graph_off"], hn, graph["pos"], P["readout.position_weights.weight"])
    index = {a: i for i, a in enumerate(cand)}
    want_ranks, per_q = [], []
    for q in queries:
        s = orc.bilinear_match(hg, ds.node_features[q].expand(len(cand), -1), P["match.W.weight"], True).squeeze(1).numpy()
        pos = [index[a] for a in ds.node2parents[q] if a in index]
        r = orc.ranks_of_positives(s, pos, True)
        want_ranks += r
        per_q.append(r)
    assert ranks.cpu().tolist() == want_ranks
    np.testing.assert_allclose(metrics["macro_mr"], np.mean([np.mean(r) for r in per_q]), rtol=1e-12)
    np.testing.assert_allclose(metrics["hit_at_3"], np.mean([np.mean(np.asarray(r) <= 3) for r in per_q]), rtol=1e-12)
    np.testing.assert_allclose(metrics["mrr_scaled_10"], np.mean([np.mean(1.0 / np.ceil(np.asarray(r) / 10)) for r in per_q]), rtol=1e-12)


def test_training_on_raw_files_with_device_batches_converges():
    """a short training run on the toy raw dataset, batches built on device (sample_anchors -> device_egonet_batch with the
    positive's query excluded), InfoNCE like trainer.py:52-56: the loss goes down (the toy embeddings are random, so only the
    training loss can be expected to move) and evaluation still runs on the trained model"""
    import os
    import random
    import shutil
    import tempfile
    import torch.nn.functional as F
    from taxoexpan_amd import TaxoExpan
    from taxoexpan_amd import graph as G
    from taxoexpan_amd.dataset import MAGDataset, MaskedGraphDataset
    from taxoexpan_amd.evaluate import evaluate
    dev = _dev()
    d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", None))
    try:
        for fn in os.listdir(os.path.join(GOLDEN_DIR, "toy_taxo")):
            shutil.copy(os.path.join(GOLDEN_DIR, "toy_taxo", fn), d)
        raw = MAGDataset("toy", d, raw=True)
        train = MaskedGraphDataset(raw, mode="train", sampling_mode=1, negative_size=7, expand_factor=20, normalize_embed=True)
        test = MaskedGraphDataset(raw, mode="test", sampling_mode=0, expand_factor=20, normalize_embed=True)
    finally:
        shutil.rmtree(d)
    random.seed(0)
    torch.manual_seed(0)
    model = TaxoExpan("PGAT", "WMR", "LBM", in_dim=8, hidden_dim=16, out_dim=16, pos_dim=4, num_layers=1, heads=[2, 1], feat_drop=0.1,
                      attn_drop=0.1, hidden_drop=0.1, out_drop=0.1).to(dev)
    before, *_ = evaluate(model, test, dev)
    dtax = train.device_taxonomy(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    feats = train.node_features.to(dev)
    losses = []
    order = list(range(len(train)))
    for step in range(150):
        idx = [order[(step * 16 + i) % len(order)] for i in range(16)]
        query, anchor, label, exclude = train.sample_anchors(idx)
        assert label.reshape(16, 8)[:, 0].tolist() == [1] * 16
        g = G.device_egonet_batch(dtax, anchor, exclude=exclude, expand_factor=train.expand_factor, seed=step)
        ids = g.ndata["_id"].cpu().numpy()
        off = g.csr(dev).graph_off.cpu().numpy()
        for i in range(0, len(anchor), 8):                       # the positive egonet never contains its own query
            assert query[i] not in ids[off[i]:off[i + 1]]
        model.train()
        opt.zero_grad()
        pred = model(g, g.ndata.pop("x"), feats[torch.from_numpy(query).to(dev)])
        loss = F.cross_entropy(pred.reshape(16, 8), torch.zeros(16, dtype=torch.long, device=dev), reduction="sum")
        loss.backward()
        opt.step()
        losses.append(float(loss))
    after, *_ = evaluate(model, test, dev)
    assert np.median(losses[-20:]) < 0.92 * np.median(losses[:20])
    assert np.isfinite(after["macro_mr"]) and after["n_queries"] == before["n_queries"] > 0


def test_device_batch_loader_builds_on_a_side_stream_what_the_inline_builder_builds():
    """data_loaders.DeviceBatchLoader (train.py's `for batch in data_loader`, batches built on the GPU inside next() on a side
    stream while the previous step runs): every batch equals the in-line construction from the same sampled anchors -- ids, pos, both
    CSR views, node and query features (handed out as ops.RepeatedRows: one row per query),
    labels -- also while the caller's stream is kept busy, and a step can consume it at once"""
    import os
    import random
    import shutil
    import tempfile
    from taxoexpan_amd import TaxoExpan
    from taxoexpan_amd import graph as G
    from taxoexpan_amd.data_loaders import DeviceBatchLoader
    from taxoexpan_amd.dataset import MAGDataset, MaskedGraphDataset
    dev = _dev()
    d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", None))
    try:
        for fn in os.listdir(os.path.join(GOLDEN_DIR, "toy_taxo")):
            shutil.copy(os.path.join(GOLDEN_DIR, "toy_taxo", fn), d)
        sets = []
        for _ in range(2):                                          # two datasets in the same sampler state
            random.seed(0)
            sets.append(MaskedGraphDataset(MAGDataset("toy", d, raw=True), mode="train", sampling_mode=1, negative_size=7, expand_factor=5,
                                           normalize_embed=True))
    finally:
        shutil.rmtree(d)
    ds_a, ds_b = sets
    torch.manual_seed(0)
    model = TaxoExpan("PGAT", "WMR", "LBM", in_dim=8, hidden_dim=16, out_dim=16, pos_dim=4, num_layers=1, heads=[2, 1], feat_drop=0.1,
                      attn_drop=0.1, hidden_drop=0.1, out_drop=0.1).to(dev).eval()
    loader = DeviceBatchLoader(ds_a, 16, dev, shuffle=True, seed=3)
    assert len(loader) == -(-len(ds_a) // 16)
    busy = torch.randn(2048, 2048, device=dev)
    host = lambda t: t.cpu().numpy().copy()
    got = []
    random.seed(1)
    for b, (g, x, qf, labels) in enumerate(loader):
        if b == 4:
            break
        busy = (busy @ busy) * 1e-3                                 # the caller's stream has work queued while next() builds
        pos = g.ndata["pos"]
        with torch.no_grad():
            pred = model(g, x, qf)                                  # consumable at once: the caller's stream waits for the build
        csr = g.csr(dev)
        from taxoexpan_amd import ops
        assert isinstance(qf, ops.RepeatedRows) and qf.rows.shape[0] == 16 and qf.shape == (16 * 8, 8)    # one distinct row per sampled query
        got.append(dict(ids=host(g.ndata["_id"]), pos=host(pos), x=host(x), qf=host(qf.dense()), labels=host(labels), pred=host(pred),
                        csr=[host(t) for t in (csr.rowptr_in, csr.col_src, csr.eid_in, csr.rowptr_out, csr.col_dst, csr.pos_out, csr.graph_off)]))
    assert len(got) == 4
    # the same four batches in line on the caller's stream, from the twin dataset
    order = list(range(len(ds_b)))
    random.Random(3).shuffle(order)
    dtax = ds_b.device_taxonomy(dev)
    random.seed(1)
    for b, want in enumerate(got):
        query, anchor, label, exclude = ds_b.sample_anchors(order[b * 16:(b + 1) * 16])
        g = G.device_egonet_batch(dtax, anchor, exclude=exclude, expand_factor=ds_b.expand_factor, seed=3 + 7919 + b)
        csr = g.csr(dev)
        assert np.array_equal(host(g.ndata["_id"]), want["ids"]) and np.array_equal(host(g.ndata["pos"]), want["pos"])
        for t, w in zip((csr.rowptr_in, csr.col_src, csr.eid_in, csr.rowptr_out, csr.col_dst, csr.pos_out, csr.graph_off), want["csr"]):
            assert np.array_equal(host(t), w)
        x = g.ndata.pop("x")
        qf = dtax.features.index_select(0, torch.from_numpy(query).to(dev))
        assert np.array_equal(host(x), want["x"]) and np.array_equal(host(qf), want["qf"]) and np.array_equal(label, want["labels"])
        with torch.no_grad():                                       # (the loader's scores come from the one-row-per-query form: another
            np.testing.assert_allclose(host(model(g, x, qf)), want["pred"], rtol=2e-5, atol=1e-7)     # summation order in V = Q W^T)
        assert np.isfinite(want["pred"]).all() and want["labels"].reshape(16, 8)[:, 0].tolist() == [1] * 16


@pytest.mark.gpu
def test_many_batches_begun_before_any_is_finished():
    """begin_device_batch / finish_device_batch with a dozen batches in flight (each job owns its pinned read-back slot, the upload
    buffers take turns behind their events): every batch equals the one-call construction of the same anchors, both query forms"""
    from taxoexpan_amd import graph as G, ops, synthetic as syn
    from taxoexpan_amd.data_loaders import begin_device_batch, build_device_batch, finish_device_batch
    dev = _dev()
    tax = syn.make_taxonomy(3000, 4500, 12, seed=3)
    dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    rs = np.random.RandomState(0)
    jobs, args = [], []
    for b in range(12):
        q = rs.randint(0, tax.n_nodes, size=16)
        anchors = rs.randint(0, tax.n_nodes, size=16 * 5)
        exclude = np.where(np.arange(80) % 5 == 0, np.repeat(q, 5), -1)
        args.append((anchors, exclude, np.repeat(q, 5)))
        jobs.append(begin_device_batch(dtax, anchors, exclude, np.repeat(q, 5), expand_factor=4, seed=b, stream=side, repeated_queries=b % 2 == 1))
    for b, (job, (anchors, exclude, qid)) in enumerate(zip(jobs, args)):
        got = finish_device_batch(job, dtax.features)
        want = build_device_batch(dtax, anchors, exclude, qid, dtax.features, expand_factor=4, seed=b)
        torch.cuda.synchronize()
        assert got["n_nodes"] == want["n_nodes"] and got["n_edges"] == want["n_edges"]
        assert torch.equal(got["g"].ndata["_id"], want["g"].ndata["_id"]) and torch.equal(got["pos"], want["pos"]) and torch.equal(got["x"], want["x"])
        ca, cb = got["g"].csr(dev), want["g"].csr(dev)
        for f in ("rowptr_in", "col_src", "eid_in", "rowptr_out", "col_dst", "pos_out", "graph_off"):
            assert torch.equal(getattr(ca, f), getattr(cb, f)), f
        qa = got["qf"]
        assert isinstance(qa, ops.RepeatedRows) == (b % 2 == 1)
        assert torch.equal(ops.dense_rows(qa), want["qf"])


@pytest.mark.gpu
def test_runs_of_stacked_query_rows_found_on_the_device_and_the_match_on_them(monkeypatch):
    """txe_rows_find_runs on the reference collate's stacked query matrix (data_loaders.py:9-28: a query's row once per pair) against
    numpy -- runs of every length, a row that returns after another (two runs), -0.0 against 0.0 (different bit patterns: two runs), more
    rows than one scan chunk, no repetition at all, one row -- and BilinearStackedRunsFunction (txe_bilinear_stacked_*) against the GEMM
    form on the same matrix: scores, d_hg, dW; then the matcher's decision and its periodic re-decision (model_zoo._Bilinear._repeats)."""
    from taxoexpan_amd import model_zoo as mz, ops
    dev = _dev()
    rs = np.random.RandomState(2)
    r, l = 250, 500
    table = rs.standard_normal((64, r)).astype(np.float32)
    table[10, :] = 0.0
    table[11, :] = 0.0
    table[11, 7] = -0.0
    for ids in (np.concatenate([[3], np.repeat([7, 1, 7, 22, 10, 11, 10], [32, 1, 5, 90, 2, 2, 1]), np.repeat(np.arange(40), 60), [5, 6]]),
                np.arange(64).repeat(1), np.array([9]), np.repeat([4], 1500)):
        e2 = torch.from_numpy(table[ids]).to(dev)
        G = len(ids)
        run_id, run_off, n_runs = ops.find_row_runs(e2)
        start = np.flatnonzero(np.concatenate([[True], ids[1:] != ids[:-1]]))
        U = len(start)
        assert int(n_runs.item()) == U
        assert run_off.cpu().numpy()[:U + 1].tolist() == start.tolist() + [G]
        assert run_id.cpu().numpy()[:G].tolist() == (np.cumsum(np.concatenate([[1], ids[1:] != ids[:-1]])) - 1).tolist()
        W = torch.from_numpy((rs.standard_normal((1, l, r)) * 0.05).astype(np.float32)).to(dev)
        e1 = torch.from_numpy((rs.standard_normal((G, l)) * 0.3).astype(np.float32)).to(dev)
        up = torch.linspace(-1, 1, G, device=dev)
        for apply_exp in (False, True):
            res = []
            for fn in (lambda a, w: ops.BilinearPairFunction.apply(a, e2, w, apply_exp, None),
                       lambda a, w: ops.BilinearStackedRunsFunction.apply(a, e2, w, apply_exp)):
                a, w = e1.clone().requires_grad_(True), W.clone().requires_grad_(True)
                sc = fn(a, w)
                (sc.reshape(-1) * up).sum().backward()
                res.append((sc.detach(), a.grad, w.grad))
            for k in range(3):
                scale = max(res[0][k].abs().max().item(), 1e-30)
                np.testing.assert_allclose(res[1][k].cpu().numpy(), res[0][k].cpu().numpy(), rtol=2e-5, atol=2e-6 * scale)
    # the matcher decides on its first training batch of at least 256 pairs and looks again every RECHECK_EVERY calls (no host sync)
    monkeypatch.setattr(ops, "_NO_QUERY_RUNS", False)
    monkeypatch.setattr(mz._Bilinear, "RECHECK_EVERY", 3)
    rep = torch.from_numpy(table[np.repeat(np.arange(16), 32)]).to(dev)
    uniq = torch.from_numpy(rs.standard_normal((512, r)).astype(np.float32)).to(dev)
    hg = torch.from_numpy(rs.standard_normal((512, l)).astype(np.float32) * 0.1).to(dev).requires_grad_(True)
    m = mz.LBM(l, r).to(dev)
    seen = []
    real = ops.BilinearStackedRunsFunction.apply
    monkeypatch.setattr(ops.BilinearStackedRunsFunction, "apply", staticmethod(lambda *a: (seen.append(1), real(*a))[1]))
    with torch.no_grad():
        m(hg, rep)
    assert "_runs_watch" not in m.__dict__ and not seen               # (no decision outside training)
    m(hg, rep).sum().backward()
    assert m._runs_watch["dec"] is True and len(seen) == 1
    m(hg, uniq).sum().backward()                                      # a batch that does not repeat: same form, still right
    a = hg.detach().clone().requires_grad_(True)
    ref = ops.BilinearPairFunction.apply(a, uniq, m.W.weight, True, None)
    np.testing.assert_allclose(m(hg, uniq).detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-5)
    assert len(seen) == 3
    # ... the loader keeps handing out rows that do not repeat: within a few batches the matcher is back on the GEMM form
    for _ in range(8):
        m(hg, uniq).sum().backward()
        torch.cuda.synchronize()
    assert m._runs_watch["dec"] is False
    n_seen = len(seen)
    m(hg, uniq).sum().backward()
    assert len(seen) == n_seen
    for _ in range(8):                                                # ... and forth again when they do
        m(hg, rep).sum().backward()
        torch.cuda.synchronize()
    assert m._runs_watch["dec"] is True
    m2 = mz.BIM(l, r).to(dev)
    n_seen = len(seen)
    m2(hg, uniq).sum().backward()
    assert m2._runs_watch["dec"] is False and len(seen) == n_seen
    m3 = mz.BIM(l, r).to(dev)
    monkeypatch.setattr(ops, "_NO_QUERY_RUNS", True)
    m3(hg, rep).sum().backward()
    assert "_runs_watch" not in m3.__dict__ and len(seen) == n_seen


@pytest.mark.gpu
@pytest.mark.parametrize("apply_exp,stacked", [(True, True), (False, False)])
def test_folded_match_kernels_against_torch_autograd(apply_exp, stacked):
    """txe_bilinear_folded_fwd / _bwd (ops.BilinearFoldedRunsFunction) on ragged runs -- lengths 1, 700 (longer than a staging pass),
    equal rows in SEPARATE runs, odd widths (D = 37, Kp = 96, r = 23) -- against the same function written in torch on float64:
    scores, dZ, the matcher's dW and the main part of the output layer's dW that travels through the FoldLink"""
    from taxoexpan_amd import ops
    dev = _dev()
    rs = np.random.RandomState(5)
    lens = [1, 700, 3, 1, 40, 255, 2]
    ids = np.repeat([4, 2, 9, 4, 1, 7, 2], lens)                                     # (ids 4 and 2 come back in later runs)
    G, D, Kp, r = len(ids), 37, 96, 23
    table = torch.from_numpy(rs.standard_normal((12, r)).astype(np.float32)).to(dev)
    Z = torch.from_numpy((0.3 * rs.standard_normal((G, Kp))).astype(np.float32)).to(dev).requires_grad_(True)
    Wp = torch.zeros(128, Kp, device=dev)
    Wp[:D] = torch.from_numpy((0.3 * rs.standard_normal((D, Kp))).astype(np.float32)).to(dev)
    Wm = torch.from_numpy((0.3 * rs.standard_normal((1, D, r))).astype(np.float32)).to(dev).requires_grad_(True)
    wts = torch.linspace(-1, 1, G, device=dev)
    link = ops.FoldLink()
    if stacked:
        e2 = table.index_select(0, torch.from_numpy(ids).to(dev))
        s = ops.BilinearFoldedRunsFunction.apply(Z, Wp, link, D, Wm, apply_exp, e2, None, None)
    else:
        rr = ops.RepeatedRows.from_ids(table, ids)
        assert rr.rows.shape[0] == len(lens)
        s = ops.BilinearFoldedRunsFunction.apply(Z, Wp, link, D, Wm, apply_exp, None, rr.rows, rr.run_off)
    (s.reshape(-1) * wts).sum().backward()
    assert link.S == 1 and tuple(link.part.shape) == (D, Kp)
    Zd = Z.detach().double().cpu().requires_grad_(True)
    Wd = Wp[:D].double().cpu().requires_grad_(True)
    Wmd = Wm.detach().double().cpu().requires_grad_(True)
    q = table.double().cpu()[torch.from_numpy(ids)]
    raw = ((Zd @ Wd.t()) * (q @ Wmd[0].t())).sum(1)
    ref = raw.exp() if apply_exp else raw
    (ref * wts.double().cpu()).sum().backward()
    tol = dict(rtol=2e-4)
    np.testing.assert_allclose(s.detach().reshape(-1).cpu().numpy(), ref.detach().numpy(), atol=2e-5 * float(ref.detach().abs().max()), **tol)
    np.testing.assert_allclose(Z.grad.cpu().numpy(), Zd.grad.numpy(), atol=2e-5 * float(Zd.grad.abs().max()), **tol)
    np.testing.assert_allclose(Wm.grad.cpu().numpy(), Wmd.grad.numpy(), atol=2e-5 * float(Wmd.grad.abs().max()), **tol)
    np.testing.assert_allclose(link.part.cpu().numpy(), Wd.grad.numpy(), atol=2e-5 * float(Wd.grad.abs().max()), **tol)


@pytest.mark.gpu
@pytest.mark.parametrize("matcher,stacked", [("LBM", True), ("BIM", True), ("LBM", False)])
def test_graph_vector_folded_into_the_matcher_equals_the_materialised_one(matcher, stacked):
    """TaxoExpan.forward on query rows that repeat: the readout stops at Z and the bilinear matcher runs the output layer's product on
    one row per query run (DeferredGraphVector -> ops.BilinearFoldedRunsFunction, txe_bilinear_folded_*) -- against the same step with hg = Z W^T formed (ops._NO_MATCH_FOLD): same dropout seeds, the loss and every
    parameter gradient within float rounding of the other association; stacked rows (runs found on the device) and ops.RepeatedRows;
    then some other consumer asks for the folded vector's tensor (FoldedGraphLinearFunction) and gets the same numbers."""
    from taxoexpan_amd import TaxoExpan, model_zoo as mz, ops, synthetic as syn
    from taxoexpan_amd.loss import info_nce_loss
    if ops._NO_MATCH_FOLD or ops._NO_FUSED_BWD or mz._NO_FOLD or (stacked and ops._NO_QUERY_RUNS):
        pytest.skip("the test route (TXE_TEST_ROUTE) switches the folded graph vector off")
    dev = _dev()
    tax = syn.make_taxonomy(900, 1400, 12, seed=6)
    nq, per = 40, 8                                                                  # 320 stacked rows: enough for the run detection
    g, qf, _labels = syn.training_batch(tax, nq, per - 1, seed=3)
    qf = qf.to(dev)
    x, pos = g.ndata.pop("x").to(dev), g.ndata["pos"].to(dev)
    host_q = qf.cpu().numpy()
    qid = np.concatenate([[0], np.cumsum(np.any(host_q[1:] != host_q[:-1], axis=1))])
    q_arg = qf if stacked else ops.RepeatedRows.from_ids(torch.from_numpy(host_q[np.concatenate([[True], qid[1:] != qid[:-1]])]).to(dev), qid)
    torch.manual_seed(2)
    model = TaxoExpan("PGAT", "WMR", matcher, in_dim=qf.shape[1], hidden_dim=16, out_dim=24, pos_dim=4, num_layers=1, heads=[4, 1], feat_drop=0.1,
                      attn_drop=0.1, hidden_drop=0.0, out_drop=0.0).to(dev).train()
    target = torch.zeros(nq, dtype=torch.long, device=dev)
    res, kinds = [], []
    prev = ops._NO_MATCH_FOLD
    try:
        for no_fold in (False, True):
            ops._NO_MATCH_FOLD = no_fold
            model.zero_grad()
            model.match.__dict__.pop("_runs_watch", None)
            g.ndata["pos"] = pos
            torch.manual_seed(7)
            with ops.debug_capture() as runs:
                loss = info_nce_loss(model(g, x, q_arg).reshape(nq, -1), target)
            loss.backward()
            kinds.append((runs[-1][1].final, dict(runs.routes)["match"], ops.ROUTES["stack_bwd"]))
            res.append((loss.item(), {n: p.grad.clone() for n, p in model.named_parameters()}, runs[-1][1].seed))
    finally:
        ops._NO_MATCH_FOLD = prev
    edot = not ops._NO_FOLD_EDOT
    assert kinds == [("collapse_z", "folded", "fused+edot" if edot else "collapse"), ("collapse", "stacked" if stacked else "runs", "collapse")], kinds
    assert res[0][2] == res[1][2]                                                    # (same dropout masks: the two steps are the same function)
    assert abs(res[0][0] - res[1][0]) <= 2e-5 * abs(res[1][0])
    for n, ref in res[1][1].items():
        np.testing.assert_allclose(res[0][1][n].cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-5 * max(ref.abs().max().item(), 1e-30), err_msg=n)
    # ---- the deferred vector under other consumers (eval mode: no dropout, so separate forwards agree) ----
    model.eval()

    def fresh():
        g.ndata["pos"] = pos
        g.ndata["h"] = model.graph_propagate(g, x)
        return model.readout(g, pos)
    with torch.no_grad():
        ref = fresh()
    assert torch.is_tensor(ref)                                                      # without gradients: computed at once
    hv = fresh()
    assert isinstance(hv, mz.DeferredGraphVector) and hv.shape == (nq * per, 24) and not hv.started() and hv.can_fold()
    # a logging hook's `.detach()`: the values, and the route stays open
    d = hv.detach()
    assert torch.is_tensor(d) and not d.requires_grad and hv.started() and hv.can_fold()
    np.testing.assert_allclose(d.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-6)
    # ... the matcher still folds afterwards (the query-side job did not ride in the sweep: the in-line form of the kernels)
    model.zero_grad()
    s1 = model.match(hv, q_arg)
    assert ops.ROUTES["match"] == "folded" and ops.ROUTES["fold"] == "inline"
    info_nce_loss(s1.reshape(nq, -1), target).backward()
    g_inline = {n: p.grad.clone() for n, p in model.named_parameters()}
    # ... against the same step with nothing touched in between (the job rides in the sweep)
    model.zero_grad()
    s2 = model.match(fresh(), q_arg)
    assert ops.ROUTES["fold"] == ("edot" if edot else "inline")
    info_nce_loss(s2.reshape(nq, -1), target).backward()
    np.testing.assert_allclose(s1.detach().cpu().numpy(), s2.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(g_inline[n].cpu().numpy(), p.grad.cpu().numpy(), rtol=2e-4, atol=2e-5 * max(p.grad.abs().max().item(), 1e-30), err_msg=n)
    # a second differentiable use AFTER the fold is refused loudly; the values stay available outside autograd
    with pytest.raises(RuntimeError, match="consumed FOLDED"):
        hv.tensor()
    with torch.no_grad():
        np.testing.assert_allclose(hv.tensor().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-6)
    # any other consumer first: the plain tensor ('collapse'), and the matcher then takes a run form on it
    hv = fresh()
    t = hv + 0.0                                                                     # a torch function: materialises
    assert torch.is_tensor(t) and not hv.can_fold() and ops.ROUTES["stack"] == "collapse"
    np.testing.assert_allclose(t.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-6)
    model.match(hv, q_arg).sum().backward()
    assert ops.ROUTES["match"] == ("stacked" if stacked else "runs")
    # .detach() and THEN a tensor consumer: hg = Z W^T as an autograd node (FoldedGraphLinearFunction), gradients as on the plain route
    model.zero_grad()
    (fresh() * 1.0).sum().backward()
    g_plain = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    hv = fresh()
    hv.detach()
    (hv * 1.0).sum().backward()
    assert ops.ROUTES["fold"] == "materialised"
    for n, ref_g in g_plain.items():
        got = dict(model.named_parameters())[n].grad
        np.testing.assert_allclose(got.cpu().numpy(), ref_g.cpu().numpy(), rtol=2e-4, atol=2e-5 * max(ref_g.abs().max().item(), 1e-30), err_msg=n)


@pytest.mark.gpu
@pytest.mark.parametrize("matcher", ["LBM", "BIM"])
def test_repeated_query_rows_match_the_stacked_rows(matcher):
    """ops.RepeatedRows as the matcher's query argument (the distinct rows of a training batch's query features + their runs,
    txe_bilinear_runs_*) against the stacked [G, r] tensor the reference's collate builds (data_loaders.py:9-28): scores, d_hg
    and dW within float rounding of the other summation order; runs of length 1, a long run, repeated ids in separate runs; the
    whole model's loss and parameter gradients on a training batch; matchers without a runs form densify it."""
    from taxoexpan_amd import TaxoExpan, model_zoo as mz, ops, synthetic as syn
    from taxoexpan_amd.loss import info_nce_loss
    dev = _dev()
    rs = np.random.RandomState(11)
    table = torch.from_numpy(rs.standard_normal((40, 250)).astype(np.float32)).to(dev)
    ids = np.concatenate([[3], np.repeat([7, 1, 7, 22], [32, 1, 5, 90]), [5, 6]])
    G, l = len(ids), 500
    rr = ops.RepeatedRows.from_ids(table, ids)
    assert rr.rows.shape[0] == 7 and rr.run_off.cpu().tolist() == [0, 1, 33, 34, 39, 129, 130, 131]
    dense = table.index_select(0, torch.from_numpy(ids).to(dev))
    assert torch.equal(rr.dense(), dense)
    m = getattr(mz, matcher)(l, 250).to(dev)
    with torch.no_grad():
        m.W.weight.mul_(0.05)
    e1 = torch.from_numpy(rs.standard_normal((G, l)).astype(np.float32) * 0.3).to(dev)
    outs = []
    for q in (dense, rr):
        a = e1.clone().requires_grad_(True)
        m.zero_grad()
        s = m(a, q)
        (s.reshape(-1) * torch.linspace(-1, 1, G, device=dev)).sum().backward()
        outs.append((s.detach().clone(), a.grad.clone(), m.W.weight.grad.clone()))
    for k in (0, 1, 2):
        scale = outs[0][k].abs().max().item()
        np.testing.assert_allclose(outs[1][k].cpu().numpy(), outs[0][k].cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)
    # ... and both run forms (RepeatedRows, and the stacked rows whose runs the device finds) against the ORACLE's bilinear match
    # (model_zoo.py:301-328 restated), not only against the library's own GEMM form
    ac, Wc = e1.cpu().clone().requires_grad_(True), m.W.weight.detach().cpu().clone().requires_grad_(True)
    sr = orc.bilinear_match(ac, dense.cpu(), Wc, matcher == "LBM")
    (sr.reshape(-1) * torch.linspace(-1, 1, G)).sum().backward()
    a2 = e1.clone().requires_grad_(True)
    m.zero_grad()
    s2 = ops.BilinearStackedRunsFunction.apply(a2, dense, m.W.weight, matcher == "LBM")
    (s2.reshape(-1) * torch.linspace(-1, 1, G, device=dev)).sum().backward()
    for got in (outs[1], (s2.detach(), a2.grad, m.W.weight.grad)):
        for g_, w_ in zip(got, (sr.detach().reshape(-1, 1), ac.grad, Wc.grad)):
            np.testing.assert_allclose(g_.cpu().numpy().reshape(w_.shape), w_.numpy(), rtol=1e-4, atol=2e-6 * float(w_.abs().max()))
    # an empty batch, and a matcher without a runs form
    assert m(e1[:0], ops.RepeatedRows.from_ids(table, np.zeros(0, dtype=np.int64))).shape == (0, 1)
    mlp = mz.MLP(l, 250, 16).to(dev)
    assert torch.equal(mlp(e1, rr), mlp(e1, dense))
    # the whole training step
    tax = syn.make_taxonomy(600, 900, 12, seed=5)
    g, qf, labels = syn.training_batch(tax, 24, 7, seed=9)
    qf = qf.to(dev)
    host_q = qf.cpu().numpy()
    qid = np.concatenate([[0], np.cumsum(np.any(host_q[1:] != host_q[:-1], axis=1))])
    rq = ops.RepeatedRows.from_ids(torch.from_numpy(host_q[np.concatenate([[True], qid[1:] != qid[:-1]])]).to(dev), qid)
    assert rq.rows.shape[0] == 24 and torch.equal(rq.dense(), qf)
    x, pos = g.ndata.pop("x").to(dev), g.ndata["pos"].to(dev)
    torch.manual_seed(1)
    model = TaxoExpan("PGAT", "WMR", matcher, in_dim=qf.shape[1], hidden_dim=16, out_dim=16, pos_dim=4, num_layers=1, heads=[2, 1], feat_drop=0.0,
                      attn_drop=0.0, hidden_drop=0.0, out_drop=0.0).to(dev)
    res = []
    for q in (qf, rq):
        model.zero_grad()
        g.ndata["pos"] = pos
        loss = info_nce_loss(model(g, x, q).reshape(24, -1), torch.zeros(24, dtype=torch.long, device=dev))
        loss.backward()
        res.append((loss.item(), {n: p.grad.clone() for n, p in model.named_parameters()}))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[0][0])
    for n in res[0][1]:
        ref = res[0][1][n]
        # (the RepeatedRows step takes the graph vector folded into the matcher, the 168 stacked rows -- too few for the run detection --
        #  the materialised one: the same sums in another association, hence the fp32-rounding atol)
        np.testing.assert_allclose(res[1][1][n].cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5 * max(ref.abs().max().item(), 1e-30), err_msg=n)


@pytest.mark.gpu
@pytest.mark.parametrize("amsgrad,wd", [(True, 0.0), (False, 0.0), (True, 0.01)])
def test_adam_step_equals_torch_adam(amsgrad, wd):
    """txe_adam_step against torch.optim.Adam (the optimizer of config.mag.json:66-73) over several steps: odd sizes, a parameter
    without gradient in some steps (its step count lags), state_dict round trip into the torch class"""
    from taxoexpan_amd.optim import Adam
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    shapes = [(3, 1), (3, 50), (1, 4, 500), (2000, 300), (1001,), (1, 500, 250), (7, 9, 11)]
    mine = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    o1 = Adam(mine, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, amsgrad=amsgrad)
    o2 = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, amsgrad=amsgrad)
    for it in range(6):
        for a, b in zip(mine, ref):
            g = torch.randn_like(a) * (10.0 ** (it % 3 - 1))
            a.grad, b.grad = g.clone(), g.clone()
        if it in (2, 3):                       # parameter 4 skips two updates
            mine[4].grad = None
            ref[4].grad = None
        o1.step()
        o2.step()
        for a, b in zip(mine, ref):
            torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)
    assert int(o1.state[mine[4]]["step"]) == 4 and int(o1.state[mine[0]]["step"]) == 6
    sd = o1.state_dict()
    o3 = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in mine], lr=1e-2, weight_decay=wd, amsgrad=amsgrad)
    o3.load_state_dict(sd)                     # same layout as torch's
    for k in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if amsgrad else ()):
        for i in range(len(shapes)):
            torch.testing.assert_close(o3.state[o3.param_groups[0]["params"][i]][k], o2.state[ref[i]][k], rtol=2e-6, atol=1e-7)   # moments near zero cancel


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,zero_target", [(128, 32, True), (8, 257, False), (1, 1, True), (37, 64, False), (0, 5, True)])
def test_info_nce_loss_equals_torch_cross_entropy(B, C, zero_target):
    """txe_info_nce against model/loss.py:52-57 (F.cross_entropy, reduction sum): value, gradient, upstream gradient scaling,
    non-contiguous rows (the LBM scores are a column of a [G, 1] tensor reshaped)"""
    from taxoexpan_amd.loss import info_nce_loss
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 131 + C)
    base = (torch.randn(B, C + 3, device=dev) * 4.0)
    x1 = base[:, :C].detach().requires_grad_(True)          # row stride C + 3
    x2 = base[:, :C].detach().clone().requires_grad_(True)
    tgt = torch.zeros(B, dtype=torch.long, device=dev) if zero_target else torch.randint(0, C, (B,), device=dev)
    l1 = info_nce_loss(x1, None if zero_target and B % 2 == 0 else tgt)
    l2 = F.cross_entropy(x2, tgt, reduction="sum")
    torch.testing.assert_close(l1, l2, rtol=1e-5, atol=1e-5)
    (l1 * 0.5).backward()
    (l2 * 0.5).backward()
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_attention_logits_fused_into_the_previous_aggregation(drop, monkeypatch):
    """the folded output layer's logits formed in layer 0's aggregation epilogue (txe_gat_aggregate_fwd nx_*) against the separate
    sweep (txe_gat_collapse_fwd a12_ready = 0): same model, same dropout seeds, forward and gradients"""
    from taxoexpan_amd import TaxoExpan, ops
    from taxoexpan_amd.graph import BatchedDGLGraph
    dev = _dev()
    rs = np.random.RandomState(11)
    ks = rs.randint(1, 4, 40).tolist() + [1, 3]
    ms = rs.randint(0, 9, 40).tolist() + [0, 60]
    g = BatchedDGLGraph.from_egonet_shapes(ks, ms)
    N = g.number_of_nodes()
    x = torch.randn(N, 12, generator=torch.Generator().manual_seed(2)).to(dev)
    q = torch.randn(len(ks), 12, generator=torch.Generator().manual_seed(3)).to(dev)
    torch.manual_seed(9)
    model = TaxoExpan("PGAT", "WMR", "LBM", in_dim=12, hidden_dim=20, out_dim=16, pos_dim=6, num_layers=1, heads=[4, 1], feat_drop=drop,
                      attn_drop=drop).to(dev).train(drop > 0)
    pos = g.ndata["pos"].clone()
    outs = []
    for no_fuse in (False, True):
        monkeypatch.setattr(ops, "_NO_FUSED_LOGITS", no_fuse)
        model.zero_grad(set_to_none=True)
        g.ndata["pos"] = pos.clone()
        torch.manual_seed(77)                                   # same dropout seeds in both passes
        s = model(g, x, q)
        (s * torch.linspace(0.5, 1.5, s.numel(), device=dev).reshape(s.shape)).sum().backward()
        outs.append((s.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}))
    (a, ga), (b, gb) = outs
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    for k in ga:
        np.testing.assert_allclose(ga[k], gb[k], rtol=1e-3, atol=1e-5, err_msg=k)        # fp32 summation order of the logits differs


@pytest.mark.gpu
@pytest.mark.parametrize("prop", ["PGAT", "GAT", "PGCN"])
def test_eval_encode_on_table_rows_equals_materialised_features(prop, monkeypatch):
    """SURVEY 8f-2 'dedup by _id': device-built egonets whose features stay rows of the taxonomy table (ops.GatheredRows) -- the
    eval-mode layer-0 projection is formed once per taxonomy node and gathered -- against the same batch with gathered features;
    chunked with the projection cache; training mode falls back to the ordinary path"""
    from taxoexpan_amd import TaxoExpan, ops, synthetic as syn, graph as G
    from taxoexpan_amd.scoring import encode_candidates
    dev = _dev()
    tax = syn.make_taxonomy(600, 900, 12, seed=4)
    dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
    torch.manual_seed(3)
    kw = dict(in_dim=12, hidden_dim=20, out_dim=16, num_layers=1, heads=[4, 1], feat_drop=0.2, attn_drop=0.2, hidden_drop=0.2, out_drop=0.0)
    if prop != "GAT":
        kw["pos_dim"] = 6
    model = TaxoExpan(prop, "WMR" if prop == "PGAT" else "MR", "LBM", **kw).to(dev).eval()
    cand = np.arange(600, dtype=np.int64)                       # every node twice: more batch nodes than table rows
    chunks = [np.concatenate([cand, cand])[i:i + 500] for i in range(0, 1200, 500)]
    lazy = [G.device_egonet_batch(dtax, c, seed=5, with_features="lazy") for c in chunks]
    full = [G.device_egonet_batch(dtax, c, seed=5) for c in chunks]
    assert isinstance(lazy[0].ndata["x"], ops.GatheredRows)
    calls = []
    name = "_gcn_table_projection" if prop == "PGCN" else "_gat_table_projection"
    orig = getattr(ops, name)
    monkeypatch.setattr(ops, name, lambda st, src: (calls.append(1), orig(st, src))[1])
    hg_l = encode_candidates(model, lazy)
    hg_f = encode_candidates(model, full)
    assert len(calls) == len(chunks)                            # the table path ran (its projection is cached after the first chunk)
    np.testing.assert_allclose(hg_l.cpu().numpy(), hg_f.cpu().numpy(), rtol=1e-4, atol=1e-5)
    # PGAT forms the projected rows inside the message/reduce sweep; materialising them first (A/B switch) is the same arithmetic
    monkeypatch.setattr(ops, "_NO_TABLE_SWEEP", True)
    assert torch.equal(encode_candidates(model, lazy), hg_l)
    monkeypatch.setattr(ops, "_NO_TABLE_SWEEP", False)
    calls.clear()
    # the lazy features are an ordinary tensor for every other consumer
    x = lazy[0].ndata["x"]
    assert tuple(x.shape) == tuple(full[0].ndata["x"].shape) and torch.equal(x + 0, full[0].ndata["x"])
    # training mode (dropout) and the A/B switch use the materialised path, same results as gathered features given the seed
    calls.clear()
    model.train()
    outs = []
    for g in (lazy[0], full[0]):
        pos = g.ndata["pos"]
        torch.manual_seed(1)
        g.ndata["h"] = model.graph_propagate(g, g.ndata["x"])
        outs.append(model.readout(g, pos).detach().cpu().numpy())
        g.ndata["pos"] = pos
    assert not calls
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("H,D,with_nx", [(4, 500, True), (4, 160, True), (4, 24, False), (2, 100, True), (1, 36, False)])
def test_table_rows_formed_inside_the_sweep_equal_materialised_rows(H, D, with_nx):
    """txe_gat_aggregate_table_fwd against txe_gather_add_rows + txe_gat_aggregate_fwd, bit for bit (same operation order), on a
    multigraph with a hub of in-degree 150 (the chunked softmax), nodes without in-edges, every row-width template, with and
    without the next layer's folded logits in the epilogue"""
    from taxoexpan_amd import _lib, graph as G
    dev = _dev()
    rs = np.random.RandomState(H * 100 + D)
    N, n_tab, vocab = 700, 90, 3
    F, Fe = H * D, H * D + 2 * H
    Fp = -(-Fe // 128) * 128
    kp = -(-(F + 6) // 32) * 32
    src = rs.randint(0, N, 2200)
    dst = rs.randint(5, N, 2200)                                                    # nodes 0..4: no in-edges
    src[:150], dst[:150] = rs.randint(0, N, 150), 17                                # a hub
    rin, col = G.build_csr_device(torch.from_numpy(src.astype(np.int32)).to(dev), torch.from_numpy(dst.astype(np.int32)).to(dev), N)[:2]
    T = torch.from_numpy(rs.standard_normal((n_tab, Fp)).astype(np.float32)).to(dev)
    T2 = torch.from_numpy(rs.standard_normal((vocab, Fp)).astype(np.float32)).to(dev)
    rid = torch.from_numpy(rs.randint(0, n_tab, N).astype(np.int32)).to(dev)
    pos = torch.from_numpy(rs.randint(0, vocab, N).astype(np.int32)).to(dev)
    wa = torch.from_numpy(rs.standard_normal((2, kp)).astype(np.float32)).to(dev)
    assert _lib.call("txe_gat_aggregate_table_supported", H, D, Fp, vocab, kp if with_nx else 0) == 1
    ld_out = kp if with_nx else F
    outs = []
    for table in (False, True):
        out = torch.full((N, ld_out), 0.25, device=dev)                              # (the columns behind F belong to the caller)
        a12 = torch.zeros(N, 2, device=dev)
        nx = (wa.data_ptr(), kp) if with_nx else (None, 0)
        if table:
            _lib.call("txe_gat_aggregate_table_fwd", rin.data_ptr(), col.data_ptr(), N, T.data_ptr(), Fp, rid.data_ptr(), T2.data_ptr(),
                      pos.data_ptr(), vocab, H, D, 0.2, 1, 0.1, out.data_ptr(), ld_out, nx[0], nx[1], a12.data_ptr() if with_nx else None,
                      1, _lib.stream_ptr())
        else:
            Y = torch.empty(N, Fp, device=dev)
            _lib.call("txe_gather_add_rows", T.data_ptr(), Fp, rid.data_ptr(), T2.data_ptr(), Fp, pos.data_ptr(), N, Fp, Y.data_ptr(), Fp,
                      _lib.stream_ptr())
            _lib.call("txe_gat_aggregate_fwd", rin.data_ptr(), col.data_ptr(), N, Y.data_ptr(), Fp, Y.data_ptr() + 4 * F,
                      Y.data_ptr() + 4 * (F + H), Fp, H, D, 0.2, 0.0, 0, 1, 0.1, out.data_ptr(), ld_out, None, nx[0], nx[1], None, 0.0,
                      a12.data_ptr() if with_nx else None, 0, _lib.stream_ptr())
        torch.cuda.synchronize()
        outs.append((out.cpu(), a12.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[1][0][:5, :F].abs().max()) == 0.0 and float(outs[1][0][17, :F].abs().max()) > 0.0
    # a shape the LDS cannot hold is refused (the caller then materialises the rows), so is H > 4
    assert _lib.call("txe_gat_aggregate_table_supported", 4, 500, 2048, 8, 2080) == 0
    assert _lib.call("txe_gat_aggregate_table_supported", 8, 64, 640, 3, 0) == 0


@pytest.mark.gpu
def test_adam_step_many_tensors_and_empty_ones():
    """more tensors than one launch's argument table holds (24), zero-element parameters, a single element"""
    from taxoexpan_amd.optim import Adam
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    shapes = [(i % 7 + 1, (i * 13) % 11) for i in range(40)] + [(1,), (0,), (1025,)]
    mine = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    o1, o2 = Adam(mine, lr=3e-3, amsgrad=True), torch.optim.Adam(ref, lr=3e-3, amsgrad=True)
    for _ in range(3):
        for a, b in zip(mine, ref):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        o1.step()
        o2.step()
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)
