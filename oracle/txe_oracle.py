"""TEST INFRASTRUCTURE ONLY -- CPU restatement (the *oracle*) of TaxoExpan's hot path.

What this is: a plain-torch, CPU, explicit-COO restatement of the arithmetic of
  /root/reference/model/model_zoo.py   (GATLayer/PGAT/GAT, GCNLayer/PGCN/GCN, readouts, matchers)
  /root/reference/model/model.py:70-87 (TaxoExpan.forward)
  /root/reference/test_fast.py:25-28,116-123 (encode_graph + per-query scoring loop)
  /root/reference/model/metric.py:7-60 (rank definition)
  /root/reference/model/loss.py:52-57 + trainer/trainer.py:52-56 (InfoNCE grouping)
  /root/reference/data_loader/dataset.py:404-437 (egonet layout)
Every function cites the reference lines it follows.  It is written with differentiable
torch ops so that torch.autograd on CPU yields the oracle gradients (the reference itself
relies on autograd + DGL's edge_softmax backward).

Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- as the
checker / the timed CPU baseline only.  The product (taxoexpan_amd/) never imports it.

How it is pinned: tests/golden/*.npz were captured by oracle/gen_golden.py from the
UNMODIFIED reference model_zoo.py imported in the build container; tests/test_oracle_golden.py
checks this file against them (<=1e-6 fp32 on CPU).  The torch arithmetic of model_zoo.py is
therefore pinned by the reference itself.  PARITY-UNPINNED part: DGL 0.4 is an un-vendored
dependency (README.md:7-13 pins "DGL 0.4.0"); the semantics of edge_softmax / update_all /
mean_nodes / batch used to run the reference come from DGL's published behaviour, restated in
oracle/dgl_shim (see its header), not from a file under /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# Egonet layout  (dataset.py:404-437)
# ----------------------------------------------------------------------------------------
def egonet_edges(k, m):
    """One egonet with k grand-parents (pos 0), the anchor (pos 1), m siblings (pos 2).

    dataset.py:429-435: node order [gp_0..gp_{k-1}, anchor, sib_0..sib_{m-1}]; edges in id
    order: k x (gp_i -> anchor), m x (anchor -> sib_j), then n self loops.  E = 2n-1.
    """
    n = k + 1 + m
    src = list(range(k)) + [k] * m + list(range(n))
    dst = [k] * k + list(range(k + 1, n)) + list(range(n))
    pos = [0] * k + [1] + [2] * m
    return n, src, dst, pos


def batch_egonets(shapes):
    """dgl.batch restated (data_loaders.py:25): concatenate, offset node ids.

    shapes: iterable of (k, m).  Returns dict(src,dst,pos: int64 tensors, graph_off: int64[G+1],
    num_nodes).
    """
    srcs, dsts, poss, off, goff = [], [], [], 0, [0]
    for (k, m) in shapes:
        n, s, d, p = egonet_edges(int(k), int(m))
        srcs.append(np.asarray(s, dtype=np.int64) + off)
        dsts.append(np.asarray(d, dtype=np.int64) + off)
        poss.append(np.asarray(p, dtype=np.int64))
        off += n
        goff.append(off)
    cat = lambda xs: torch.from_numpy(np.concatenate(xs)) if xs else torch.zeros(0, dtype=torch.long)
    return dict(src=cat(srcs), dst=cat(dsts), pos=cat(poss),
                graph_off=torch.tensor(goff, dtype=torch.long), num_nodes=off)


# ----------------------------------------------------------------------------------------
# DGL primitives on explicit COO  [DGL 0.4 published semantics -- parity-unpinned]
# ----------------------------------------------------------------------------------------
def edge_softmax(dst, n, logits):
    """softmax over incoming edges of every destination (model_zoo.py:112)."""
    tail = tuple(logits.shape[1:])
    idx = dst.reshape((-1,) + (1,) * len(tail)).expand_as(logits)
    mx = torch.full((n,) + tail, float("-inf"), dtype=logits.dtype)
    mx = mx.scatter_reduce(0, idx, logits.detach(), reduce="amax", include_self=True)
    ex = torch.exp(logits - mx[dst])
    den = torch.zeros((n,) + tail, dtype=logits.dtype).index_add(0, dst, ex)
    return ex / den[dst]


def scatter_sum(dst, n, msg):
    """update_all(..., fn.sum): out[v] = sum of messages on edges ending in v; 0 for no in-edges."""
    return torch.zeros((n,) + tuple(msg.shape[1:]), dtype=msg.dtype).index_add(0, dst, msg)


def segment_sum(graph_off, x):
    counts = graph_off[1:] - graph_off[:-1]
    gid = torch.repeat_interleave(torch.arange(len(counts)), counts)
    return torch.zeros((len(counts),) + tuple(x.shape[1:]), dtype=x.dtype).index_add(0, gid, x)


# ----------------------------------------------------------------------------------------
# GAT  (model_zoo.py:52-114, 169-220)
# ----------------------------------------------------------------------------------------
BRANCH_AUDIT = None      # a list while a test audits given branches (see _leaky)


def _leaky(x, slope, pos=None, care=None, tag=""):
    """F.leaky_relu; with `pos` (bool, same shape) the branch of every element is GIVEN instead of decided by its sign.  leaky_relu' is
    discontinuous at 0: two fp32 evaluations whose pre-activation differs in the last bit around 0 differentiate different linear
    pieces.  A test that hands the implementation-under-test's own branch pattern to the oracle compares gradients of the SAME
    piecewise-linear function, entry for entry (the forward value moves by at most |x|(1-slope) ~ one rounding error there).
    The given pattern is AUDITED when BRANCH_AUDIT is a list: a record (tag, entries that disagree with the sign of the oracle's own
    pre-activation, largest |x| among them, largest |x| overall) per call -- a given branch may differ from the oracle's own only where
    |x| is within rounding of 0.  care (bool, broadcastable): entries that carry a gradient at all (the rest is not audited: an entry
    the next layer's dropout zeroed reads as "negative" on the device whatever its sign was)."""
    if pos is None:
        return F.leaky_relu(x, slope)
    if BRANCH_AUDIT is not None:
        with torch.no_grad():
            dis = (x > 0) != pos
            if care is not None:
                dis = dis & care
            ax = x.abs()
            BRANCH_AUDIT.append((tag, int(dis.sum()), float(ax[dis].max()) if bool(dis.any()) else 0.0, float(ax.max()), int(x.numel())))
    return torch.where(pos, x, x * slope)


def gat_layer(src, dst, n, feature, fc_w, attn_l, attn_r, slope=0.2, feat_keep=None, attn_keep=None,
              feat_scale=1.0, attn_scale=1.0, return_parts=False, residual=False, res_w=None, e_pos=None):
    """GATLayer.forward, model_zoo.py:80-104 (residual branch :98-103 is dead for PGAT: `residual=True` adds
    res_fc(h) -- or h itself broadcast over heads when res_w is None, i.e. in_dim == out_dim).

    feature N x K; fc_w (H*D) x K; attn_l/attn_r 1 x H x D.  Dropout is expressed through explicit
    keep masks (feat_keep N x K, attn_keep E x H x 1, values 0/1) and their 1/(1-p) scales.
    Returns N x H x D.
    """
    H = attn_l.shape[1]
    h = feature if feat_keep is None else feature * feat_keep * feat_scale        # :82
    ft = (h @ fc_w.t()).reshape(h.shape[0], H, -1)                                  # :83
    a1 = (ft * attn_l).sum(-1, keepdim=True)                                         # :84
    a2 = (ft * attn_r).sum(-1, keepdim=True)                                         # :85
    e = _leaky(a1[src] + a2[dst], slope, e_pos, tag="attention logits")             # :106-109 (e_pos: given branches, see _leaky)
    alpha = edge_softmax(dst, n, e)                                                  # :111-112
    a_drop = alpha if attn_keep is None else alpha * attn_keep * attn_scale          # :114
    out = scatter_sum(dst, n, ft[src] * a_drop)                                      # :95
    if residual:                                                                     # :98-103
        out = out + ((h @ res_w.t()).reshape(h.shape[0], H, -1) if res_w is not None else h.unsqueeze(1))
    if return_parts:
        return out, dict(ft=ft, a1=a1, a2=a2, e=e, alpha=alpha)
    return out


def pgat_forward(params, graph, h, heads, num_layers, act_slope=0.01, attn_slope=0.2, prefix="",
                 masks=None, positional=True, return_parts=False):
    """PGAT.forward (model_zoo.py:210-220) / GAT.forward (:183-190 when positional=False).

    params: state-dict-keyed tensors: gat_layers.{i}.fc.weight / .attn_l / .attn_r,
    prop_position_embeddings.{i}.weight.  activation = F.leaky_relu (slope 0.01, model.py:40).
    masks: optional list (one per GATLayer) of dict(feat_keep, attn_keep, feat_scale, attn_scale).
    """
    src, dst, pos, n = graph["src"], graph["dst"], graph["pos"], graph["num_nodes"]
    parts = []
    for l in range(num_layers + 1):
        w = params[f"{prefix}gat_layers.{l}.fc.weight"]
        al = params[f"{prefix}gat_layers.{l}.attn_l"]
        ar = params[f"{prefix}gat_layers.{l}.attn_r"]
        x = h
        if positional:
            p = params[f"{prefix}prop_position_embeddings.{l}.weight"][pos]        # :214 / :218
            x = torch.cat((h, p), 1)                                                 # :215 / :219
        mk = dict((masks[l] if masks is not None else None) or {})
        act_pos = mk.pop("act_pos", None)                                            # (given branches of the activation below)
        out, pr = gat_layer(src, dst, n, x, w, al, ar, attn_slope, return_parts=True, **mk)
        parts.append(pr)
        if l < num_layers:
            nk = (masks[l + 1] or {}).get("feat_keep") if (masks is not None and act_pos is not None) else None
            care = (nk[:, :out.shape[1] * out.shape[2]] > 0) if nk is not None else None   # (dropped next-layer inputs carry no gradient)
            h = _leaky(out.flatten(1), act_slope, act_pos, care, tag=f"activation after layer {l}")   # :215-216
        else:
            h = out.mean(1)                                                          # :219
    if return_parts:
        return h, parts
    return h


# ----------------------------------------------------------------------------------------
# GCN  (model_zoo.py:13-50, 116-167)
# ----------------------------------------------------------------------------------------
def gcn_norm(dst, n, dtype=torch.float32):
    """in_degree^-1/2 with inf -> 0 (model_zoo.py:157-161); in-degree counts self loops."""
    deg = torch.bincount(dst, minlength=n).to(dtype)
    norm = torch.pow(deg, -0.5)
    norm[torch.isinf(norm)] = 0
    return norm.unsqueeze(1)


def gcn_layer(src, dst, n, h, weight, bias, norm, act_slope=None, keep=None, keep_scale=1.0, act_pos=None, act_care=None):
    """GCNLayer.forward, model_zoo.py:34-50."""
    if keep is not None:
        h = h * keep * keep_scale                                                     # :35-36
    h = h @ weight                                                                    # :37
    h = h * norm                                                                      # :39
    h = scatter_sum(dst, n, h[src])                                                   # :41
    h = h * norm                                                                      # :44
    if bias is not None:
        h = h + bias                                                                  # :47
    if act_slope is not None:
        h = _leaky(h, act_slope, act_pos, act_care, tag="GCN activation")             # :49
    return h


def pgcn_forward(params, graph, h, num_layers, act_slope=0.01, prefix="", masks=None, positional=True):
    """PGCN.forward (model_zoo.py:155-167) / GCN.forward (:128-137).  Last layer has no activation
    (:126,:152)."""
    src, dst, pos, n = graph["src"], graph["dst"], graph["pos"], graph["num_nodes"]
    norm = gcn_norm(dst, n, h.dtype)
    for l in range(num_layers + 1):
        x = h
        if positional:
            x = torch.cat((h, params[f"{prefix}prop_position_embeddings.{l}.weight"][pos]), 1)  # :165-166
        mk = dict((masks[l] if masks is not None else None) or {})
        if mk.get("act_pos") is not None and l < num_layers and (masks[l + 1] or {}).get("keep") is not None:
            mk["act_care"] = masks[l + 1]["keep"][:, :params[f"{prefix}layers.{l}.weight"].shape[1]] > 0   # (dropped next-layer inputs carry no gradient)
        h = gcn_layer(src, dst, n, x, params[f"{prefix}layers.{l}.weight"], params.get(f"{prefix}layers.{l}.bias"),
                      norm, act_slope if l < num_layers else None, **mk)
    return h


# ----------------------------------------------------------------------------------------
# Readouts (model_zoo.py:227-276)
# ----------------------------------------------------------------------------------------
def mean_readout(graph_off, h):
    """MeanReadout: dgl.mean_nodes(g,'h') (model_zoo.py:231-232)."""
    cnt = (graph_off[1:] - graph_off[:-1]).to(h.dtype).unsqueeze(1)
    return segment_sum(graph_off, h) / cnt


def weighted_mean_readout(graph_off, h, pos, position_weights):
    """WeightedMeanReadout (model_zoo.py:240-242): w = softplus(Emb(3,1)[pos]);
    hg = segsum(w*h)/segsum(w)."""
    w = F.softplus(position_weights[pos])                     # N x 1
    return segment_sum(graph_off, h * w) / segment_sum(graph_off, w)


def sum_readout(graph_off, h):
    """SumReadout (model_zoo.py:260-267)."""
    return segment_sum(graph_off, h)


def max_readout(graph_off, h):
    """MaxReadout (model_zoo.py:269-276)."""
    return torch.stack([h[int(a):int(b)].max(0)[0] for a, b in zip(graph_off[:-1], graph_off[1:])])


def concat_readout(graph_off, h, pos):
    """ConcatReadout (model_zoo.py:248-258)."""
    cnt = (graph_off[1:] - graph_off[:-1]).to(h.dtype).unsqueeze(1)
    a_gp = (pos == 0).to(h.dtype).unsqueeze(1)
    a_p = (pos == 1).to(h.dtype).unsqueeze(1)
    a_sib = (pos == 2).to(h.dtype).unsqueeze(1)
    gp = segment_sum(graph_off, h * a_gp) / cnt
    p = segment_sum(graph_off, h * a_p) / segment_sum(graph_off, a_p)
    sib = segment_sum(graph_off, h * a_sib) / cnt
    return torch.cat((gp, p, sib), 1)


# ----------------------------------------------------------------------------------------
# Matchers (model_zoo.py:281-346)
# ----------------------------------------------------------------------------------------
def bilinear_match(e1, e2, W, apply_exp):
    """BIM (:313) / LBM (:328): s_i = e1_i^T W[0] e2_i ; LBM returns exp(s).  W: 1 x l x r."""
    s = torch.einsum("il,lr,ir->i", e1, W[0], e2).unsqueeze(1)
    return torch.exp(s) if apply_exp else s


def mlp_match(e1, e2, w0, b0, w1, b1):
    """MLP (:285-298)."""
    return F.linear(F.relu(F.linear(torch.cat((e1, e2), 1), w0, b0)), w1, b1)


def ntn_match(e1, e2, u_w, W_w, W_b, V_w, non_linear=torch.tanh):
    """NTN (:331-346): u_R(f(Bilinear_k(e1, e2) + bias + V(cat(e1, e2))))."""
    bil = torch.einsum("gl,klr,gr->gk", e1, W_w, e2) + W_b
    return non_linear(bil + torch.cat((e1, e2), 1) @ V_w.t()) @ u_w.t()


# ----------------------------------------------------------------------------------------
# Model assembly, loss, scoring loop, ranks
# ----------------------------------------------------------------------------------------
def taxoexpan_forward(params, graph, h, qf, propagation="PGAT", readout="WMR", matching="LBM", heads=(4, 1),
                      num_layers=1, masks=None):
    """TaxoExpan.forward (model.py:70-87) with state-dict-keyed params
    (graph_propagate.*, readout.position_weights.weight, match.W.weight)."""
    if propagation in ("PGAT", "GAT"):
        hn = pgat_forward(params, graph, h, heads, num_layers, prefix="graph_propagate.", masks=masks,
                          positional=(propagation == "PGAT"))
    else:
        hn = pgcn_forward(params, graph, h, num_layers, prefix="graph_propagate.", masks=masks,
                          positional=(propagation == "PGCN"))
    if readout == "WMR":
        hg = weighted_mean_readout(graph["graph_off"], hn, graph["pos"], params["readout.position_weights.weight"])
    elif readout == "MR":
        hg = mean_readout(graph["graph_off"], hn)
    else:
        hg = concat_readout(graph["graph_off"], hn, graph["pos"])
    scores = bilinear_match(hg, qf, params["match.W.weight"], apply_exp=(matching == "LBM"))
    return scores, hg, hn


def info_nce_loss(scores, n_queries):
    """trainer.py:52-56 + loss.py:52-57: reshape (n_queries, 1+neg), CE against class 0, sum."""
    pred = scores.reshape(n_queries, -1)
    return F.cross_entropy(pred, torch.zeros(n_queries, dtype=torch.long), reduction="sum")


def score_all_literal(hg, W, queries, apply_exp):
    """The per-query loop of test_fast.py:116-123: for every query expand to G rows and call
    model.match(hg, expanded).  Returns Q x G."""
    rows = []
    for q in queries:
        rows.append(bilinear_match(hg, q.expand(hg.shape[0], -1), W, apply_exp).squeeze(1))
    return torch.stack(rows)


def ranks_of_positives(scores_row, positive_idx, larger_is_better=True):
    """metric.py:7-31 on one query: rank = 1 + #negatives STRICTLY better than the positive
    (positives are masked out of the comparison set)."""
    s = np.asarray(scores_row)
    pos = np.asarray(positive_idx, dtype=np.int64)
    neg_mask = np.ones(len(s), dtype=bool)
    neg_mask[pos] = False
    neg = s[neg_mask]
    if larger_is_better:
        return [int((neg > s[p]).sum()) + 1 for p in pos]
    return [int((neg < s[p]).sum()) + 1 for p in pos]


def xavier_normal_std(fan_in, fan_out, gain=1.414):
    return gain * math.sqrt(2.0 / (fan_in + fan_out))
