"""TEST INFRASTRUCTURE ONLY -- a CPU, pure-torch stand-in for the DGL 0.4 surface.

Purpose: let the UNMODIFIED reference file /root/reference/model/model_zoo.py be
imported in the build container (DGL 0.4 is not installed and cannot be) so that
oracle/gen_golden.py can capture golden vectors from the reference's own Python.
Nothing in the product (taxoexpan_amd/) imports this package.

PARITY-UNPINNED: the semantics of the DGL primitives restated here come from the
published DGL 0.4.x behaviour (README.md:7-13 of the reference pins "DGL 0.4.0"),
not from a file under /root/reference -- DGL is an un-vendored third-party
dependency.  Restated primitives and the reference call sites that use them:

  edge_softmax(g, logits)            model_zoo.py:6,112   softmax over the incoming
                                                           edges of each destination node
  update_all(src_mul_edge, sum)      model_zoo.py:95      out[v] = sum_{e=(u->v)} ft[u]*a[e]
  update_all(copy_src, sum)          model_zoo.py:41      out[v] = sum_{e=(u->v)} h[u]
  apply_edges(udf)                   model_zoo.py:90,108  udf sees edges.src / edges.dst views
  in_degrees()                       model_zoo.py:130,157 counts self loops
  mean_nodes / sum_nodes             model_zoo.py:232,242,252-256
                                      mean_nodes(g,f,w) = segsum(w*f)/segsum(w)
  batch(list)                        data_loaders.py:25   concatenate, offset ids
  DGLGraph().add_nodes/add_edges     dataset.py:429-435
  g.to_networkx()                    dataset.py:225       DiGraph, nodes 0..n-1, edges in edge-id order
"""
import torch

from . import function  # noqa: F401


class _Frame(dict):
    """ndata / edata: a dict of tensors with pop(); DGL frames behave like this
    for every use the reference makes of them."""


class _EdgeBatch:
    def __init__(self, g):
        self.src = {k: v[g._src] for k, v in g.ndata.items() if torch.is_tensor(v) and v.shape[0] == g._n}
        self.dst = {k: v[g._dst] for k, v in g.ndata.items() if torch.is_tensor(v) and v.shape[0] == g._n}
        self.data = g.edata


class DGLGraph:
    def __init__(self):
        self._n = 0
        self._src = torch.zeros(0, dtype=torch.long)
        self._dst = torch.zeros(0, dtype=torch.long)
        self.ndata = _Frame()
        self.edata = _Frame()

    # -- construction (dataset.py:429-435) ------------------------------------
    def add_nodes(self, num, data=None):
        assert self._n == 0 or not data, "shim: features only on first add_nodes"
        self._n += int(num)
        if data:
            for k, v in data.items():
                self.ndata[k] = v

    def add_edges(self, u, v):
        u = torch.as_tensor(u, dtype=torch.long).reshape(-1)
        v = torch.as_tensor(v, dtype=torch.long).reshape(-1)
        if u.numel() == 0 or v.numel() == 0:
            return
        if u.numel() == 1 and v.numel() > 1:
            u = u.expand(v.numel())
        if v.numel() == 1 and u.numel() > 1:
            v = v.expand(u.numel())
        assert u.numel() == v.numel()
        self._src = torch.cat([self._src, u])
        self._dst = torch.cat([self._dst, v])

    def nodes(self):
        return torch.arange(self._n)

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.numel())

    def edges(self):
        return self._src, self._dst

    def in_degrees(self):
        return torch.bincount(self._dst, minlength=self._n)

    def to_networkx(self):
        """dataset.py:225 -- DGL 0.4: nx.DiGraph, nodes 0..n-1, edges added in edge-id order (attribute 'id')"""
        import networkx as nx
        g = nx.DiGraph()
        g.add_nodes_from(range(self._n))
        for eid, (u, v) in enumerate(zip(self._src.tolist(), self._dst.tolist())):
            g.add_edge(u, v, id=eid)
        return g

    # -- message passing --------------------------------------------------------
    def apply_edges(self, udf):
        out = udf(_EdgeBatch(self))
        for k, v in out.items():
            self.edata[k] = v

    def update_all(self, msg, red):
        src, dst = self._src, self._dst
        if msg.kind == "copy_src":
            m = self.ndata[msg.src][src]
        elif msg.kind == "src_mul_edge":
            m = self.ndata[msg.src][src] * self.edata[msg.edge]
        else:  # pragma: no cover
            raise NotImplementedError(msg.kind)
        assert red.kind == "sum" and red.msg == msg.out
        out = torch.zeros((self._n,) + tuple(m.shape[1:]), dtype=m.dtype, device=m.device)
        out = out.index_add(0, dst.to(m.device), m)
        self.ndata[red.out] = out


class BatchedDGLGraph(DGLGraph):
    def __init__(self, graphs):
        super().__init__()
        self.batch_size = len(graphs)
        self.batch_num_nodes = [g.number_of_nodes() for g in graphs]
        self.batch_num_edges = [g.number_of_edges() for g in graphs]
        srcs, dsts, off = [], [], 0
        for g in graphs:
            srcs.append(g._src + off)
            dsts.append(g._dst + off)
            off += g._n
        self._n = off
        self._src = torch.cat(srcs) if srcs else torch.zeros(0, dtype=torch.long)
        self._dst = torch.cat(dsts) if dsts else torch.zeros(0, dtype=torch.long)
        if graphs:
            for k in graphs[0].ndata.keys():
                self.ndata[k] = torch.cat([g.ndata[k] for g in graphs], 0)


def batch(graphs):
    return BatchedDGLGraph(list(graphs))


def _graph_ids(g, device):
    counts = torch.as_tensor(g.batch_num_nodes, dtype=torch.long)
    return torch.repeat_interleave(torch.arange(len(counts)), counts).to(device)


def _seg_sum(g, x):
    gid = _graph_ids(g, x.device)
    out = torch.zeros((len(g.batch_num_nodes),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    return out.index_add(0, gid, x)


def _bw(w, x):
    """DGL 0.4 readout reshapes a per-node weight to (N,1,..,1) so that an (N,) or (N,1) weight
    broadcasts over the feature dims (model_zoo.py:251-256 passes an (N,) weight)."""
    return w.reshape((-1,) + (1,) * (x.dim() - 1))


def sum_nodes(g, feat, weight=None):
    x = g.ndata[feat]
    if weight is not None:
        x = x * _bw(g.ndata[weight], x)
    return _seg_sum(g, x)


def mean_nodes(g, feat, weight=None):
    x = g.ndata[feat]
    if weight is not None:
        w = _bw(g.ndata[weight], x)
        return _seg_sum(g, x * w) / _seg_sum(g, w)
    n = torch.as_tensor(g.batch_num_nodes, dtype=x.dtype, device=x.device).unsqueeze(1)
    return _seg_sum(g, x) / n


def max_nodes(g, feat):
    x = g.ndata[feat]
    outs, off = [], 0
    for n in g.batch_num_nodes:
        outs.append(x[off:off + n].max(0)[0])
        off += n
    return torch.stack(outs)
