"""CPU: the oracle restatement (oracle/txe_oracle.py) against the golden vectors captured from the
UNMODIFIED reference (oracle/gen_golden.py).  fp32 on CPU both sides -> tight tolerances."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import txe_oracle as orc
from golden_util import GOLDEN_DIR, check_grad, load_case, oracle_masks

RT, AT = 2e-5, 2e-6


@pytest.mark.parametrize("name", [n for n, s in gc.CASES.items() if s["match"] != "MLP"])
def test_oracle_matches_reference_goldens(name):
    spec, z, shapes, x, q, params, graph = load_case(name)
    P = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    masks = oracle_masks(spec, params, graph)
    scores, hg, hn = orc.taxoexpan_forward(P, graph, torch.from_numpy(x), torch.from_numpy(q), spec["prop"],
                                           spec["readout"], spec["match"], spec["heads"], spec["num_layers"], masks)
    loss = orc.info_nce_loss(scores, spec["n_queries"])
    loss.backward()
    np.testing.assert_allclose(hn.detach().numpy()[::gc.row_steps(spec)[0]], z["hn"], rtol=RT, atol=AT)
    np.testing.assert_allclose(hg.detach().numpy(), z["hg"], rtol=RT, atol=AT)
    np.testing.assert_allclose(scores.detach().numpy(), z["scores"], rtol=RT, atol=AT)
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-5)
    for k, p in P.items():
        check_grad(z, k, p.grad.numpy(), rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("name", ["small_pgat_wmr_lbm", "small_pgat_2layer", "small_pgat_dropout", "mag_pgat_wmr_lbm"])
def test_oracle_gat_intermediates(name):
    spec, z, shapes, x, q, params, graph = load_case(name)
    P = {k: torch.from_numpy(v) for k, v in params.items()}
    masks = oracle_masks(spec, params, graph)
    hn, parts = orc.pgat_forward(P, graph, torch.from_numpy(x), spec["heads"], spec["num_layers"],
                                 prefix="graph_propagate.", masks=masks, return_parts=True)
    for l, pr in enumerate(parts):
        alpha = pr["alpha"]
        if masks is not None and "attn_keep" in masks[l]:
            alpha = alpha * masks[l]["attn_keep"] * masks[l]["attn_scale"]     # reference stores a_drop (:114)
        np.testing.assert_allclose(alpha.numpy(), z[f"layer{l}_alpha"], rtol=RT, atol=AT)
        # attention rows sum to one over each destination's in-edges
        s = torch.zeros(graph["num_nodes"], alpha.shape[1], 1).index_add(0, graph["dst"], pr["alpha"])
        np.testing.assert_allclose(s.numpy(), 1.0, rtol=1e-5)


def test_oracle_cr_mlp_case():
    spec, z, shapes, x, q, params, graph = load_case("small_pgat_cr_mlp")
    P = {k: torch.from_numpy(v) for k, v in params.items()}
    hn = orc.pgat_forward(P, graph, torch.from_numpy(x), spec["heads"], spec["num_layers"], prefix="graph_propagate.")
    hg = orc.concat_readout(graph["graph_off"], hn, graph["pos"])
    s = orc.mlp_match(hg, torch.from_numpy(q), P["match.ffn.0.weight"], P["match.ffn.0.bias"],
                      P["match.ffn.2.weight"], P["match.ffn.2.bias"])
    np.testing.assert_allclose(hg.numpy(), z["hg"], rtol=RT, atol=AT)
    np.testing.assert_allclose(s.numpy(), z["scores"], rtol=RT, atol=AT)


def test_oracle_scoring_loop_and_ranks():
    z = dict(np.load(f"{GOLDEN_DIR}/scoring.npz"))
    hg, qs, W, positives = gc.make_scoring_inputs()
    for kind, ex in (("lbm", True), ("bim", False)):
        S = orc.score_all_literal(torch.from_numpy(hg), torch.from_numpy(W), torch.from_numpy(qs), ex).numpy()
        np.testing.assert_allclose(S, z[f"S_{kind}"], rtol=2e-5, atol=1e-6)
        ranks = []
        for qi in range(S.shape[0]):
            # metric.py is evaluated on the reference's own scores so the rank definition is tested exactly
            ranks += orc.ranks_of_positives(z[f"S_{kind}"][qi], positives[qi], larger_is_better=True)
        assert ranks == z[f"ranks_{kind}"].tolist()


def test_newterm_magnitudes_scores_and_top5_tie_order():
    """infer.py:23-38,96-106 on new-term vectors of data/mag_cs_new637.txt's magnitude (rows divided by their SUM, entries up to
    ~270): the oracle's LBM / BIM scores against the reference's -- same infs where exp overflows, same zeros where it underflows --
    and the top-5 by Python's stable sort, both directions, against scoring.topk_parents on the reference's own score rows (equal infs /
    equal zeros / duplicated candidates come out in candidate order)"""
    from taxoexpan_amd.scoring import topk_parents
    z = dict(np.load(f"{GOLDEN_DIR}/newterms.npz"))
    hg, raw, W = gc.make_newterm_inputs()
    nf32 = (raw / raw.sum(axis=1)[:, None]).astype(np.float32)
    assert np.array_equal(nf32, z["nf32"]) and np.abs(nf32).max() > 100
    ids = torch.arange(hg.shape[0])
    for kind, ex in (("lbm", True), ("bim", False)):
        S = orc.score_all_literal(torch.from_numpy(hg), torch.from_numpy(W), torch.from_numpy(nf32), ex).numpy()
        ref = z[f"S_{kind}"]
        fin = np.isfinite(ref) & (ref != 0)
        assert np.array_equal(np.isinf(S), np.isinf(ref)) and np.array_equal(S == 0, ref == 0)
        np.testing.assert_allclose(S[fin], ref[fin], rtol=5e-5)
        for larger, key in ((True, "desc"), (False, "asc")):
            got = topk_parents(torch.from_numpy(ref), ids, 5, larger).numpy()
            assert np.array_equal(got, z[f"top5_{key}_{kind}"]), (kind, key)
    assert np.isinf(z["S_lbm"]).sum() > 50 and (z["S_lbm"] == 0).sum() > 50       # the case does exercise overflow and underflow


def test_egonet_layout():
    n, s, d, p = orc.egonet_edges(2, 3)
    assert n == 6 and len(s) == 2 * n - 1
    assert s == [0, 1, 2, 2, 2, 0, 1, 2, 3, 4, 5] and d == [2, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5]
    assert p == [0, 0, 1, 2, 2, 2]
    n, s, d, p = orc.egonet_edges(0, 0)
    assert (n, s, d, p) == (1, [0], [0], [1])


def _extras():
    import os
    return np.load(os.path.join(GOLDEN_DIR, "extras.npz"))


def test_oracle_ntn_and_residual_gat_against_reference_goldens():
    """modules model/model.py never instantiates (NTN :331-346, GATLayer residual :98-103), pinned by oracle/gen_golden.py"""
    z = _extras()
    t = lambda k: torch.from_numpy(z[k]).clone().requires_grad_(True)
    e1, e2 = t("ntn.e1"), t("ntn.e2")
    P = {k: t("ntn.p." + k) for k in ("u_R.weight", "W.weight", "W.bias", "V.weight")}
    out = orc.ntn_match(e1, e2, P["u_R.weight"], P["W.weight"], P["W.bias"], P["V.weight"])
    (out * torch.from_numpy(z["ntn.coef"])).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), z["ntn.out"], rtol=RT, atol=AT)
    np.testing.assert_allclose(e1.grad.numpy(), z["ntn.d_e1"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(e2.grad.numpy(), z["ntn.d_e2"], rtol=2e-4, atol=2e-6)
    for k, p in P.items():
        np.testing.assert_allclose(p.grad.numpy(), z["ntn.g." + k], rtol=2e-4, atol=2e-6)
    graph = orc.batch_egonets(gc.EDGE_SHAPES)
    for tag in ("res_fc", "res_id"):
        x = t(tag + ".x")
        P = {k: t(f"{tag}.p.{k}") for k in ("fc.weight", "attn_l", "attn_r")}
        rw = t(tag + ".p.res_fc.weight") if tag == "res_fc" else None
        out = orc.gat_layer(graph["src"], graph["dst"], graph["num_nodes"], x, P["fc.weight"], P["attn_l"], P["attn_r"], residual=True, res_w=rw)
        (out * torch.from_numpy(z[tag + ".coef"])).sum().backward()
        np.testing.assert_allclose(out.detach().numpy(), z[tag + ".out"], rtol=RT, atol=AT)
        np.testing.assert_allclose(x.grad.numpy(), z[tag + ".d_x"], rtol=2e-4, atol=2e-6)
        for k, p in P.items():
            np.testing.assert_allclose(p.grad.numpy(), z[f"{tag}.g.{k}"], rtol=2e-4, atol=2e-6)
        if rw is not None:
            np.testing.assert_allclose(rw.grad.numpy(), z[tag + ".g.res_fc.weight"], rtol=2e-4, atol=2e-6)
