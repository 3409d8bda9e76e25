"""GPU: the routes a step can take through model_zoo -- which form of the bilinear match, whether the graph vector is folded into it,
how the propagation stack ends -- over a seeded sample of the whole space:

    grad mode on / off  x  queries stacked (the reference collate, data_loaders.py:9-28) / ops.RepeatedRows / rows that never repeat
    x  a readout hook that does nothing / logs `out.detach()` / touches the tensor (`out + 0`)  x  G in {200, 256, 4096} egonets
    x  dropout off / 0.1 (the oracle gets the kernels' own counter-based masks)  x  num_layers 1 / 2
    x  PGAT / GAT / PGCN / GCN  x  MeanReadout / WeightedMeanReadout  x  BIM / LBM  x  head layouts with and without a foldable output layer

For every sample: the route that RAN (ops.ROUTES, noted by the code itself) must be the route `_expected` derives from the inputs alone,
and scores, loss and every parameter gradient must agree with the CPU oracle (reference: model/model.py:70-87, model_zoo.py:301-328,
trainer/trainer.py:52-56).  A route change that nobody asked for -- the kind a harmless-looking hook once caused -- fails here."""
import os

import numpy as np
import pytest
import torch

import txe_oracle as orc

pytestmark = pytest.mark.gpu


WIDE = os.environ.get("TXE_ROUTE_FUZZ_WIDE", "0") == "1"     # on demand: also draw MAG / SemEval-sized widths (minutes of CPU oracle)


def _samples(n=28, seed=20260929):
    rs = np.random.RandomState(seed)
    out, seen = [], set()
    # the corners that matter are always in; the rest is drawn
    fixed = [dict(prop="PGAT", readout="WMR", match="LBM", heads=[4, 1], hidden=16, grad=True, queries="stacked", hook=None, G=4096),
             dict(prop="PGAT", readout="WMR", match="BIM", heads=[4, 1], hidden=16, grad=True, queries="stacked", hook="detach", G=256),
             dict(prop="PGAT", readout="MR", match="LBM", heads=[2, 1], hidden=16, grad=True, queries="rows", hook="touch", G=256),
             dict(prop="PGAT", readout="WMR", match="LBM", heads=[4, 1], hidden=16, grad=True, queries="stacked", hook=None, G=200),
             dict(prop="PGAT", readout="WMR", match="LBM", heads=[4, 1], hidden=10, grad=True, queries="stacked", hook=None, G=256),
             dict(prop="PGCN", readout="MR", match="BIM", heads=None, hidden=16, grad=True, queries="stacked", hook=None, G=256),
             dict(prop="GAT", readout="MR", match="BIM", heads=[2, 2], hidden=16, grad=True, queries="rows", hook=None, G=256),
             dict(prop="PGAT", readout="WMR", match="LBM", heads=[4, 1], hidden=16, grad=False, queries="rows", hook="detach", G=256)]
    for c in fixed:
        out.append(c)
        seen.add(repr(sorted(c.items(), key=lambda kv: kv[0])))
    while len(out) < n:
        prop = str(rs.choice(["PGAT", "GAT", "PGCN", "GCN"], p=[0.5, 0.15, 0.25, 0.1]))
        c = dict(prop=prop, readout=str(rs.choice(["MR", "WMR"])), match=str(rs.choice(["BIM", "LBM"])),
                 heads=([[4, 1], [2, 1], [1, 1], [2, 2]][rs.randint(4)] if prop in ("PGAT", "GAT") else None),
                 hidden=int(rs.choice([16, 16, 10])), grad=bool(rs.rand() < 0.8), queries=str(rs.choice(["stacked", "rows", "unique"])),
                 hook=[None, "detach", "touch"][rs.randint(3)], G=int(rs.choice([200, 256, 256, 4096])))
        if rs.rand() < 0.3:
            c["drop"] = 0.1                                           # feature / attention dropout on, the oracle gets the kernels' own masks
        if WIDE:                                                     # realistic widths too: vector / tile remainders of every kernel
            c.update(in_dim=int(rs.choice([12, 50, 128, 250, 300])), out_dim=int(rs.choice([24, 100, 300, 500])),
                     pos_dim=int(rs.choice([4, 10, 50])), hidden=int(rs.choice([16, 10, 64, 100, 126, 250, 500, 600])),
                     G=int(rs.choice([200, 256, 256, 512])))
            if c["heads"] is not None and rs.rand() < 0.3:
                c["heads"] = [3, 1]
        if rs.rand() < 0.2:
            c["layers"] = 2                                           # num_layers = 2: three GAT / GCN layers (heads [a, a, last])
            if c["heads"] is not None:
                c["heads"] = [c["heads"][0]] + list(c["heads"])
        key = repr(sorted(c.items(), key=lambda kv: kv[0]))
        if key not in seen:
            seen.add(key)
            out.append(c)
    return out


# (a wider sweep on demand: TXE_ROUTE_FUZZ_N=300 TXE_ROUTE_FUZZ_SEED=7 python -m pytest tests/test_gpu_routes.py -m gpu)
SAMPLES = _samples(int(os.environ.get("TXE_ROUTE_FUZZ_N", "28")), int(os.environ.get("TXE_ROUTE_FUZZ_SEED", "20260929")))


def _id(c):
    return "-".join(str(c[k]).replace(" ", "") for k in ("prop", "readout", "match", "heads", "hidden", "grad", "queries", "hook", "G") +
                    (("in_dim", "out_dim", "pos_dim") if "in_dim" in c else ())) + ("-drop" if c.get("drop") else "") + ("-L2" if c.get("layers") == 2 else "")


def _expected(c, n_nodes):
    """(match route, stack route, fold kind or None, stack_bwd route or None) from the inputs and the library's switches alone"""
    from taxoexpan_amd import _lib, model_zoo as mz, ops
    gat = c["prop"] in ("PGAT", "GAT")
    H = c["heads"]
    pd = c.get("pos_dim", 4) if c["prop"] in ("PGAT", "PGCN") else 0
    deferred_nodes = not mz._NO_FOLD and (not gat or H[-1] == 1)     # graph_propagate returns a DeferredNodeOutput
    lazy = c["grad"] and deferred_nodes                               # ... and the readout, in grad mode, a DeferredGraphVector
    if c["queries"] == "rows":
        runs = "runs"
    elif c["queries"] == "stacked" and c["grad"] and c["G"] >= 256 and not ops._NO_QUERY_RUNS:
        runs = "stacked"
    else:
        runs = None
    if gat:
        foldable = (lazy and c["hook"] != "touch" and not (ops._NO_MATCH_FOLD or ops._NO_FUSED_BWD) and
                    _lib.call("txe_gat_fused_bwd_supported", H[-2] * c["hidden"], pd, H[-2], c["hidden"]) == 1)
    else:       # a GCN stack folds when its output layer's input width leaves a padding column for the bias row (hidden + pd never a multiple of 32 here)
        foldable = lazy and c["hook"] != "touch" and not ops._NO_MATCH_FOLD and (c["hidden"] + pd) % 32 != 0
    folded = bool(runs and c["grad"] and foldable)
    match = "folded" if folded else (runs or "pair")
    edot = (folded and gat and c["hook"] != "detach" and not ops._NO_FOLD_EDOT and
            _lib.call("txe_gat_collapse_e_tiles", n_nodes, c["G"], H[-2] * c["hidden"], pd) > 0)
    if folded:          # (a GCN stack has no sweep for the matcher's job to ride in: always the in-line kernels)
        stack, fold = "collapse_z" + ("+edot" if edot else ""), ("edot" if edot else "inline")
    elif deferred_nodes:
        stack, fold = "collapse", None
    else:
        stack, fold = ("mean" if gat else "layers"), None
    # (a detach() on a foldable vector runs the stack as 'collapse_z' even when the matcher then takes another form: hg = Z W^T is
    #  materialised as an autograd node afterwards)
    if not folded and lazy and c["hook"] == "detach" and foldable:
        stack, fold = "collapse_z", "materialised"
    if not c["grad"]:
        bwd = None
    elif gat:
        bwd = "fused+edot" if edot else ("collapse" if deferred_nodes else "layers")
    else:
        bwd = None
    return match, stack, fold, bwd


@pytest.mark.parametrize("c", SAMPLES, ids=[_id(c) for c in SAMPLES])
def test_the_expected_route_runs_and_agrees_with_the_oracle(c, monkeypatch):
    from taxoexpan_amd import TaxoExpan, ops, synthetic as syn
    dev = torch.device("cuda:0")
    per = 25 if c["G"] == 200 else 32
    nq = c["G"] // per
    in_dim, out_dim, pos_dim = c.get("in_dim", 12), c.get("out_dim", 24), c.get("pos_dim", 4)
    tax = syn.make_taxonomy(3000, 4700, in_dim, seed=8)
    g, qf, _labels = syn.training_batch(tax, nq, per - 1, seed=5 + c["G"])
    if c["queries"] == "unique":                                     # one distinct row per pair: nothing repeats
        gen = torch.Generator().manual_seed(3)
        qf = torch.nn.functional.normalize(torch.randn(qf.shape, generator=gen), dim=1)
    x = g.ndata.pop("x")
    pos = g.ndata["pos"].clone()
    gat = c["prop"] in ("PGAT", "GAT")
    L = int(c.get("layers", 1))
    torch.manual_seed(11)
    drop = float(c.get("drop", 0.0))
    model = TaxoExpan(c["prop"], c["readout"], c["match"], in_dim=in_dim, hidden_dim=c["hidden"], out_dim=out_dim, pos_dim=pos_dim, num_layers=L,
                      heads=c["heads"], feat_drop=drop, attn_drop=drop, hidden_drop=drop, out_drop=drop).to(dev).train()
    seed = 1234567 + c["G"]
    monkeypatch.setattr(ops, "new_seed", lambda: seed)               # (the kernels' counter-based masks: taxoexpan_amd/rng.py restates them)
    with torch.no_grad():
        model.match.W.weight.mul_(3.0)                               # (spread the scores: InfoNCE rows that are not flat)
    q_dev = qf.to(dev)
    if c["queries"] == "rows":
        q_arg = ops.RepeatedRows(q_dev[::per].contiguous(), torch.arange(0, c["G"] + 1, per, dtype=torch.int32, device=dev), c["G"])
        assert torch.equal(q_arg.dense(), q_dev)
    else:
        q_arg = q_dev
    seen = []
    if c["hook"] == "detach":                                        # a logging hook (what oracle/gen_golden.py does to the reference)
        def hook(_m, _i, o):
            seen.append(o.detach())
        model.readout.register_forward_hook(hook)
    elif c["hook"] == "touch":                                       # a consumer of the tensor itself
        def hook(_m, _i, o):
            seen.append((o + 0.0).shape)
        model.readout.register_forward_hook(hook)
    ops.ROUTES.clear()
    with torch.set_grad_enabled(c["grad"]), ops.debug_capture() as runs:
        scores = model(g, x.to(dev), q_arg)
    want = _expected(c, int(x.shape[0]))
    taken = dict(runs.routes)
    # (with the 'detach' hook on a vector the matcher does not fold, the LAST 'fold' note is the materialisation)
    got = (taken.get("match"), taken.get("stack"), taken.get("fold"))
    assert got == want[:3], (got, want)
    target = torch.zeros(nq, dtype=torch.long, device=dev)
    branches = None
    if c["grad"]:
        # the branch every leaky_relu took on the device goes to the oracle (audited below): both sides differentiate the same
        # piecewise-linear function, so a pre-activation within rounding of 0 cannot decide a comparison (test_gpu_full_size.py)
        from test_gpu_full_size import _device_branches
        assert len(runs) == 1
        _csr_dev, _cfg, states = runs[0]
        branches = _device_branches("PGAT" if gat else "PGCN", states, np.asarray(g._src), np.asarray(g._dst), [{} for _ in states])
        loss = torch.nn.functional.cross_entropy(scores.reshape(nq, -1), target, reduction="sum")
        loss.backward()
        torch.cuda.synchronize()
        assert ops.ROUTES.get("stack_bwd") == want[3], (ops.ROUTES, want)
    # ---- the oracle ----
    csr = g.csr("cpu")
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    graph = dict(src=torch.from_numpy(np.asarray(g._src)).long(), dst=torch.from_numpy(np.asarray(g._dst)).long(), pos=pos.long(),
                 graph_off=csr.graph_off.long(), num_nodes=csr.n_nodes)
    masks = branches
    if drop > 0.0:
        from test_gpu_full_size import _masks
        masks = _masks("PGAT" if gat else "PGCN", P, c["heads"], L, csr.n_nodes, csr.n_edges, seed, csr.eid_in.numpy(), drop, drop)
        for mk, br in zip(masks, branches or [{} for _ in masks]):
            mk.update(br)
    orc.BRANCH_AUDIT = [] if branches is not None else None
    try:
        s_ref, _hg, _hn = orc.taxoexpan_forward(P, graph, x, qf, c["prop"], c["readout"], c["match"], c["heads"], L, masks)
        audit = list(orc.BRANCH_AUDIT or [])
    finally:
        orc.BRANCH_AUDIT = None
    for tag, n_dis, worst, biggest, numel in audit:                  # given branches differ from the oracle's own only within rounding of 0
        assert worst <= 1e-4 * biggest and n_dis <= 1e-3 * numel + 1, (tag, n_dis, worst, biggest, numel)
    sr = s_ref.detach().numpy()
    np.testing.assert_allclose(scores.detach().cpu().numpy(), sr, rtol=1e-4, atol=2e-5 * float(np.abs(sr).max()))
    if c["grad"]:
        l_ref = orc.info_nce_loss(s_ref, nq)
        l_ref.backward()
        np.testing.assert_allclose(loss.item(), l_ref.item(), rtol=1e-4)
        gscale = max(float(P[k].grad.abs().max()) for k, _p in model.named_parameters())
        for k, p in model.named_parameters():
            ref = P[k].grad.numpy()
            # (+ an absolute term for gradients that are exactly 0 in exact arithmetic -- the output layer's bias under InfoNCE (it shifts
            #  every score of a query alike), a one-head output layer's attn_r when all logits of a destination sit on one side of the
            #  leaky_relu (the softmax is shift-invariant; float64 oracle: 1e-14): both sides hold rounding noise there, proportional to
            #  the step's gradient scale and to sqrt(G).  Wide sweeps found 2.7e-6 at G = 4,096, and 4e-7 of the largest gradient on EACH side of a 2e-4-sized attn_r -- the device nearer to float64 than the oracle)
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3,
                                       atol=3e-4 * float(np.abs(ref).max()) + 2e-6 * max(1.0, (c["G"] / 256.0) ** 0.5) + 1e-6 * gscale, err_msg=k)
