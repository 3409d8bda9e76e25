"""Seeded synthetic taxonomies of the reference's dataset shapes + vectorised egonet batching.

No dataset can be downloaded here, and the reference's pickled datasets embed DGL-0.4 objects, so benchmarks and
tests run on random-embedding taxonomies with the published shapes:
    MAG-CS   29,654 nodes / 46,248 edges, d=250      MAG-Full 431,416 / 698,743, d=250      SemEval 83,073, d=300
A taxonomy is a DAG in topological order (node 0 = root); every other node draws 1+Poisson(lam) distinct parents among
EARLIER nodes with a heavy-tailed (Zipf-like) preference, which yields ~83-88 % leaves like the real data.
Egonets follow data_loader/dataset.py:404-437 exactly (parents of the anchor, the anchor, <= expand_factor children
sampled WITH replacement when there are more), built for a whole batch at once with numpy (no per-egonet objects).
"""
import numpy as np
import torch

from .graph import BatchedDGLGraph

SHAPES = {
    "mag_cs": dict(n_nodes=29654, n_edges=46248, dim=250),
    "mag_full": dict(n_nodes=431416, n_edges=698743, dim=250),
    "semeval_noun": dict(n_nodes=83073, n_edges=86000, dim=300),
}


class Taxonomy:
    """parent/child CSR of a DAG + L2-normalised random embeddings (normalize_embed=true, dataset.py:222-223)."""

    def __init__(self, n_nodes, par_ptr, par_idx, chd_ptr, chd_idx, features):
        self.n_nodes = n_nodes
        self.par_ptr, self.par_idx = par_ptr, par_idx      # parents of node v: par_idx[par_ptr[v]:par_ptr[v+1]]
        self.chd_ptr, self.chd_idx = chd_ptr, chd_idx
        self.features = features                           # torch float32 [n_nodes, dim] (host)

    @property
    def n_edges(self):
        return int(self.par_idx.size)

    def leaves(self):
        return np.nonzero(np.diff(self.chd_ptr) == 0)[0]


def make_taxonomy(n_nodes, n_edges, dim, seed=47, zipf=1.15, features=True):
    rs = np.random.RandomState(seed)
    lam = max(n_edges / max(n_nodes - 1, 1) - 1.0, 0.0)
    n_par = 1 + rs.poisson(lam, size=n_nodes)
    n_par[0] = 0
    n_par = np.minimum(n_par, np.arange(n_nodes))          # at most i distinct earlier nodes
    child = np.repeat(np.arange(n_nodes), n_par)
    w = (np.arange(n_nodes) + 1.0) ** (-zipf)
    cw = np.cumsum(w)
    u = rs.uniform(size=child.size) * cw[child - 1]        # parents are drawn among nodes < child
    parent = np.searchsorted(cw, u, side="right").astype(np.int64)
    parent = np.minimum(parent, child - 1)
    pairs = np.unique(np.stack([parent, child], 1), axis=0)   # distinct parents
    parent, child = pairs[:, 0], pairs[:, 1]
    order = np.argsort(child, kind="stable")
    par_idx = parent[order]
    par_ptr = np.concatenate([[0], np.cumsum(np.bincount(child, minlength=n_nodes))])
    order_c = np.argsort(parent, kind="stable")
    chd_idx = child[order_c]
    chd_ptr = np.concatenate([[0], np.cumsum(np.bincount(parent, minlength=n_nodes))])
    feats = None
    if features:
        g = torch.Generator().manual_seed(seed)
        feats = torch.randn(n_nodes, dim, generator=g)
        feats = torch.nn.functional.normalize(feats, p=2, dim=1)
    return Taxonomy(n_nodes, par_ptr, par_idx, chd_ptr, chd_idx, feats)


def make_named_taxonomy(name, seed=47, features=True):
    s = SHAPES[name]
    return make_taxonomy(s["n_nodes"], s["n_edges"], s["dim"], seed=seed, features=features)


def split_candidates(tax, seed=47):
    """10 % of the leaves -> validation, 10 % -> test, the rest of the nodes = candidate anchors (dataset.py:173-179,251)."""
    rs = np.random.RandomState(seed)
    leaves = tax.leaves()
    leaves = leaves[leaves != 0]
    rs.shuffle(leaves)
    n_hold = int(len(leaves) * 0.1)
    val, test = leaves[:n_hold], leaves[n_hold:2 * n_hold]
    held = np.zeros(tax.n_nodes, dtype=bool)
    held[val] = True
    held[test] = True
    return np.nonzero(~held)[0], val, test


def egonet_batch(tax, anchors, expand_factor=50, seed=0, exclude_child=None, with_features=True):
    """Batch the egonets of `anchors` (dataset.py:404-437).  exclude_child[i] >= 0 removes that query node from the
    sibling set of egonet i (the positive example, instance_mode 1).  Returns a BatchedDGLGraph with ndata 'x' (host
    tensor), '_id', 'pos'."""
    rs = np.random.RandomState(seed)
    anchors = np.asarray(anchors, dtype=np.int64)
    G = anchors.size
    k = (tax.par_ptr[anchors + 1] - tax.par_ptr[anchors]).astype(np.int64)
    deg = (tax.chd_ptr[anchors + 1] - tax.chd_ptr[anchors]).astype(np.int64)
    m_raw = np.minimum(deg, expand_factor)
    # sibling slots (before exclusion): all children, or expand_factor draws with replacement (random.choices, :419)
    gid = np.repeat(np.arange(G), m_raw)
    slot = np.arange(gid.size) - np.repeat(np.concatenate([[0], np.cumsum(m_raw)])[:-1], m_raw)
    big = deg[gid] > expand_factor
    pick = np.where(big, (rs.uniform(size=gid.size) * deg[gid]).astype(np.int64), slot)
    sib = tax.chd_idx[tax.chd_ptr[anchors[gid]] + pick]
    if exclude_child is not None:
        ex = np.asarray(exclude_child, dtype=np.int64)
        keep = sib != ex[gid]
        sib, gid = sib[keep], gid[keep]
    m = np.bincount(gid, minlength=G).astype(np.int64)
    n = k + 1 + m
    noff = np.concatenate([[0], np.cumsum(n)])
    ids = np.empty(int(noff[-1]), dtype=np.int64)
    # grand-parents
    gp_g = np.repeat(np.arange(G), k)
    gp_slot = np.arange(gp_g.size) - np.repeat(np.concatenate([[0], np.cumsum(k)])[:-1], k)
    ids[noff[gp_g] + gp_slot] = tax.par_idx[tax.par_ptr[anchors[gp_g]] + gp_slot]
    ids[noff[:-1] + k] = anchors
    sib_slot = np.arange(gid.size) - np.repeat(np.concatenate([[0], np.cumsum(m)])[:-1], m)
    ids[noff[gid] + k[gid] + 1 + sib_slot] = sib
    g = BatchedDGLGraph.from_egonet_shapes(k, m)
    g.ndata["_id"] = torch.from_numpy(ids)
    if with_features and tax.features is not None:
        g.ndata["x"] = tax.features[torch.from_numpy(ids)]
    return g


def training_batch(tax, n_queries, negative_size, seed=0, expand_factor=50, candidates=None):
    """One InfoNCE batch in the trainer's layout (trainer.py:52-56): for each query 1 positive anchor (a true parent,
    query removed from its siblings) followed by `negative_size` negative anchors.  Returns (graph, query_feats,
    labels) like collate_graph_and_node_small_batch (data_loaders.py:9-28)."""
    rs = np.random.RandomState(seed)
    if candidates is None:
        candidates = np.arange(tax.n_nodes)
    has_par = np.nonzero(np.diff(tax.par_ptr) > 0)[0]
    queries = rs.choice(has_par, size=n_queries, replace=len(has_par) < n_queries)
    pos_parent = tax.par_idx[tax.par_ptr[queries] + (rs.uniform(size=n_queries) * (tax.par_ptr[queries + 1] - tax.par_ptr[queries])).astype(np.int64)]
    negs = rs.choice(candidates, size=(n_queries, negative_size))
    anchors = np.concatenate([pos_parent[:, None], negs], 1).reshape(-1)
    exclude = np.full((n_queries, 1 + negative_size), -1, dtype=np.int64)
    exclude[:, 0] = queries
    g = egonet_batch(tax, anchors, expand_factor, seed=seed + 1, exclude_child=exclude.reshape(-1))
    qf = tax.features[torch.from_numpy(np.repeat(queries, 1 + negative_size))]
    labels = torch.zeros(n_queries, 1 + negative_size, dtype=torch.long)
    labels[:, 0] = 1
    return g, qf, labels.reshape(-1)
