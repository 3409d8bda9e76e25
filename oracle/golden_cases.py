"""TEST INFRASTRUCTURE ONLY -- deterministic definitions of the golden cases.

Shared by oracle/gen_golden.py (runs in the build container, imports the reference) and by the
tests (run anywhere, never touch /root/reference).  Inputs and parameters are regenerated from
seeds with numpy's legacy RandomState (bit-stable across numpy versions), so the committed
fixtures only need to hold the reference's OUTPUTS for the large-dimension cases.
"""
import math

import numpy as np

# name -> spec.  dims follow config_files/config.mag.json / config.wordnet.json for the "mag"/"semeval" cases.
CASES = {
    "small_pgat_wmr_lbm": dict(prop="PGAT", readout="WMR", match="LBM", in_dim=12, hidden_dim=8, out_dim=7, pos_dim=5,
                               num_layers=1, heads=[3, 1], n_queries=4, seed=101, full=True),
    "small_pgat_wmr_bim": dict(prop="PGAT", readout="WMR", match="BIM", in_dim=12, hidden_dim=8, out_dim=7, pos_dim=5,
                               num_layers=1, heads=[4, 1], n_queries=4, seed=102, full=True),
    "small_pgat_2layer": dict(prop="PGAT", readout="WMR", match="LBM", in_dim=10, hidden_dim=6, out_dim=9, pos_dim=4,
                              num_layers=2, heads=[2, 4, 1], n_queries=3, seed=103, full=True),
    "small_pgat_dropout": dict(prop="PGAT", readout="WMR", match="BIM", in_dim=12, hidden_dim=8, out_dim=7, pos_dim=5,
                               num_layers=1, heads=[4, 1], n_queries=4, seed=104, full=True, dropout=(0.3, 0.25)),
    "small_pgcn_mr_bim": dict(prop="PGCN", readout="MR", match="BIM", in_dim=12, hidden_dim=8, out_dim=7, pos_dim=5,
                              num_layers=1, heads=None, n_queries=4, seed=105, full=True),
    "small_pgcn_dropout": dict(prop="PGCN", readout="WMR", match="LBM", in_dim=12, hidden_dim=8, out_dim=7, pos_dim=5,
                               num_layers=2, heads=None, n_queries=4, seed=106, full=True, dropout=(0.3, 0.0)),
    "small_gat_mr_bim": dict(prop="GAT", readout="MR", match="BIM", in_dim=12, hidden_dim=8, out_dim=7, pos_dim=5,
                             num_layers=1, heads=[2, 2], n_queries=4, seed=107, full=True),
    "small_gcn_wmr_lbm": dict(prop="GCN", readout="WMR", match="LBM", in_dim=12, hidden_dim=8, out_dim=7, pos_dim=5,
                              num_layers=1, heads=None, n_queries=4, seed=108, full=True),
    "small_pgat_cr_mlp": dict(prop="PGAT", readout="CR", match="MLP", in_dim=12, hidden_dim=8, out_dim=7, pos_dim=5,
                              num_layers=1, heads=[4, 1], n_queries=4, seed=109, full=True),
    "mag_pgat_wmr_lbm": dict(prop="PGAT", readout="WMR", match="LBM", in_dim=250, hidden_dim=500, out_dim=500,
                             pos_dim=50, num_layers=1, heads=[4, 1], n_queries=3, seed=201, full=False),
    "mag_pgat_wmr_bim": dict(prop="PGAT", readout="WMR", match="BIM", in_dim=250, hidden_dim=500, out_dim=500,
                             pos_dim=50, num_layers=1, heads=[4, 1], n_queries=3, seed=202, full=False),
    "mag_pgcn_mr_bim": dict(prop="PGCN", readout="MR", match="BIM", in_dim=250, hidden_dim=500, out_dim=500,
                            pos_dim=50, num_layers=1, heads=None, n_queries=3, seed=203, full=False),
    "semeval_pgat_wmr_lbm": dict(prop="PGAT", readout="WMR", match="LBM", in_dim=300, hidden_dim=600, out_dim=300,
                                 pos_dim=50, num_layers=1, heads=[4, 1], n_queries=3, seed=204, full=False),
    "magfull_pgat_2layer": dict(prop="PGAT", readout="WMR", match="BIM", in_dim=250, hidden_dim=500, out_dim=500,
                                pos_dim=50, num_layers=2, heads=[4, 4, 1], n_queries=2, seed=205, full=False),
    # a training batch in the trainer's own layout (trainer.py:45-56, data_loaders.py:9-28): 8 queries x (1 positive + 31 negatives) = 256
    # egonets, every query's feature row stacked 32 times -- large enough (>= 256 stacked rows that repeat) for the MI355X path to take
    # the graph vector FOLDED into the bilinear matcher, so that route is pinned to the reference itself, not only to the oracle
    "mag_pgat_wmr_lbm_q8x32": dict(prop="PGAT", readout="WMR", match="LBM", in_dim=250, hidden_dim=500, out_dim=500,
                                   pos_dim=50, num_layers=1, heads=[4, 1], n_queries=8, seed=206, full=False, neg_per_query=31,
                                   repeat_queries=True, slim=True),
    "semeval_pgat_wmr_bim_q8x32": dict(prop="PGAT", readout="WMR", match="BIM", in_dim=300, hidden_dim=600, out_dim=300,
                                       pos_dim=50, num_layers=1, heads=[4, 1], n_queries=8, seed=207, full=False, neg_per_query=31,
                                       repeat_queries=True, slim=True),
}

NEG_PER_QUERY = 5  # each query: 1 positive + 5 negative egonets (trainer.py:52-56 layout)

# Degenerate egonets every case starts with (SURVEY 8c case 5): root anchor w/o parents or children,
# leaf with one parent, the expand_factor cap (50 siblings), multi-parent anchor.
EDGE_SHAPES = [(0, 0), (1, 0), (0, 3), (3, 50), (2, 1), (4, 7)]


def row_steps(spec):
    """(stride of the stored node-state rows `hn`, stride of the stored per-layer output rows `layer{l}_out`): fixture size"""
    if spec.get("slim"):
        return 10, 40
    return 1, (1 if spec["full"] else 5)


def egonet_shapes(spec):
    rs = np.random.RandomState(spec["seed"])
    g = spec["n_queries"] * (1 + spec.get("neg_per_query", NEG_PER_QUERY))
    shapes = list(EDGE_SHAPES)
    while len(shapes) < g:
        shapes.append((int(rs.randint(0, 4)), int(rs.randint(0, 9))))
    return shapes[:g]


def param_shapes(spec):
    """state-dict key -> (shape, init kind), in the reference's own key order (model.py:22-65)."""
    p = {}
    prop, H = spec["prop"], spec["heads"]
    ind, hid, out, pd, L = spec["in_dim"], spec["hidden_dim"], spec["out_dim"], spec["pos_dim"], spec["num_layers"]
    positional = prop in ("PGAT", "PGCN")
    extra = pd if positional else 0
    if prop in ("PGAT", "GAT"):
        ins = [ind + extra] + [hid * H[l - 1] + extra for l in range(1, L)] + [hid * H[-2] + extra]
        outs = [hid] * L + [out]
        for l in range(L + 1):
            p[f"graph_propagate.gat_layers.{l}.attn_l"] = ((1, H[l], outs[l]), "xavier")
            p[f"graph_propagate.gat_layers.{l}.attn_r"] = ((1, H[l], outs[l]), "xavier")
            p[f"graph_propagate.gat_layers.{l}.fc.weight"] = ((H[l] * outs[l], ins[l]), "xavier")
    else:
        ins = [ind + extra] + [hid + extra] * L
        outs = [hid] * L + [out]
        for l in range(L + 1):
            p[f"graph_propagate.layers.{l}.weight"] = ((ins[l], outs[l]), "gcn")
            p[f"graph_propagate.layers.{l}.bias"] = ((outs[l],), "gcn")
    if positional:
        for l in range(L + 1):
            p[f"graph_propagate.prop_position_embeddings.{l}.weight"] = ((3, pd), "normal")
    l_dim = out * 3 if spec["readout"] == "CR" else out
    if spec["readout"] == "WMR":
        p["readout.position_weights.weight"] = ((3, 1), "normal")
    if spec["match"] in ("LBM", "BIM"):
        p["match.W.weight"] = ((1, l_dim, ind), "bilinear")
    else:
        p["match.ffn.0.weight"] = ((hid, l_dim + ind), "linear")
        p["match.ffn.0.bias"] = ((hid,), "linear")
        p["match.ffn.2.weight"] = ((1, hid), "linear")
        p["match.ffn.2.bias"] = ((1,), "linear")
    return p


def make_params(spec):
    rs = np.random.RandomState(spec["seed"] + 7)
    out = {}
    for k, (shape, kind) in param_shapes(spec).items():
        if kind == "xavier":
            if len(shape) == 3:
                fan_in, fan_out = shape[1] * shape[2], shape[0] * shape[2]
            else:
                fan_out, fan_in = shape
            std = 1.414 * math.sqrt(2.0 / (fan_in + fan_out))
            a = rs.standard_normal(shape) * std
        elif kind == "gcn":
            stdv = 1.0 / math.sqrt(shape[-1])
            a = rs.uniform(-stdv, stdv, size=shape)
        elif kind == "normal":
            a = rs.standard_normal(shape)
        elif kind == "bilinear":
            b = 1.0 / math.sqrt(shape[1])
            a = rs.uniform(-b, b, size=shape)
        else:
            b = 1.0 / math.sqrt(shape[-1])
            a = rs.uniform(-b, b, size=shape)
        out[k] = a.astype(np.float32)
    return out


def make_inputs(spec):
    """x rows L2-normalised like dataset.py:222-223; q rows likewise."""
    rs = np.random.RandomState(spec["seed"] + 13)
    shapes = egonet_shapes(spec)
    n = sum(k + 1 + m for k, m in shapes)
    x = rs.standard_normal((n, spec["in_dim"]))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    if spec.get("repeat_queries"):              # the collate's stack: one row per query, repeated once per (query, egonet) pair
        q = rs.standard_normal((spec["n_queries"], spec["in_dim"]))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        q = np.repeat(q, len(shapes) // spec["n_queries"], axis=0)
    else:
        q = rs.standard_normal((len(shapes), spec["in_dim"]))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    return shapes, x.astype(np.float32), q.astype(np.float32)


def make_dropout_masks(spec, layer_in_dims, n_nodes, n_edges, heads):
    """fixed keep masks per layer for the dropout cases: (feat_keep N x K, attn_keep E x H x 1)."""
    rs = np.random.RandomState(spec["seed"] + 29)
    pf, pa = spec["dropout"]
    masks = []
    for l, kin in enumerate(layer_in_dims):
        fk = (rs.uniform(size=(n_nodes, kin)) >= pf).astype(np.float32)
        ak = None
        if heads is not None:
            ak = (rs.uniform(size=(n_edges, heads[l], 1)) >= pa).astype(np.float32)
        masks.append((fk, ak))
    return masks


SCORING_CASE = dict(G=203, Q=16, l_dim=500, r_dim=250, seed=301)


def make_scoring_inputs(c=SCORING_CASE):
    rs = np.random.RandomState(c["seed"])
    hg = rs.standard_normal((c["G"], c["l_dim"])).astype(np.float32) * 0.3
    qs = rs.standard_normal((c["Q"], c["r_dim"])).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    b = 1.0 / math.sqrt(c["l_dim"])
    W = rs.uniform(-b, b, size=(1, c["l_dim"], c["r_dim"])).astype(np.float32)
    positives = [sorted(set(rs.randint(0, c["G"], size=rs.randint(1, 4)).tolist())) for _ in range(c["Q"])]
    return hg, qs, W, positives


NEWTERM_CASE = dict(G=211, Q=24, l_dim=500, r_dim=250, seed=637)


def make_newterm_inputs(c=NEWTERM_CASE):
    """new-term vectors of the magnitude of the reference's data/mag_cs_new637.txt (250 values of O(+-15), |v| up to ~46, row sums
    anywhere between -200 and +200, the smallest |row sum| of that file 0.56) BEFORE infer.py:33-35 divides every row by its SUM:
    seeded synthetic rows -- three of them pinned to a tiny, a tiny negative and a large row sum -- candidate vectors hg and a
    bilinear weight.  Returns (hg [G,l], raw [Q,r] float64, W [1,l,r])."""
    rs = np.random.RandomState(c["seed"])
    raw = np.clip(rs.standard_normal((c["Q"], c["r_dim"])) * 8.0, -46.0, 46.0)
    raw[0, -1] += 0.56 - raw[0].sum()               # the file's smallest row sum: entries grow ~80x
    raw[1, -1] += -0.9 - raw[1].sum()               # a negative sum flips every sign
    raw[2, -1] += 206.7 - raw[2].sum()              # the file's largest
    hg = (rs.standard_normal((c["G"], c["l_dim"])) * 0.5).astype(np.float32)
    hg[7] = hg[3]                                   # duplicate candidates: exactly equal scores -> the order of ties is visible
    hg[150] = hg[3]
    W = (rs.standard_normal((1, c["l_dim"], c["r_dim"])) * 0.05).astype(np.float32)
    return hg, raw, W
