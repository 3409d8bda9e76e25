"""Drop-in `model_zoo` for MI355X: same class names, constructor / forward signatures, parameter names and shapes
(state-dict compatible) as /root/reference/model/model_zoo.py -- the arithmetic runs in the HIP kernels of
libtxe (include/txe.h) instead of DGL + torch.

    reference class                      here
    GCNLayer   model_zoo.py:13-50        GCNLayer   -> txe_gcn_project_* + txe_gcn_aggregate_*
    GATLayer   model_zoo.py:52-114       GATLayer   -> txe_gat_project_* + txe_gat_aggregate_*
    GCN/PGCN   model_zoo.py:116-167      GCN/PGCN   -> one fused GCNStackFunction over all layers
    GAT/PGAT   model_zoo.py:169-220      GAT/PGAT   -> one fused GATStackFunction over all layers
    MeanReadout/WeightedMeanReadout      model_zoo.py:227-242 -> txe_readout_*
    ConcatReadout/SumReadout/MaxReadout  model_zoo.py:244-276 -> txe_readout_multi_*
    MLP        model_zoo.py:281-298      MLP        -> txe_linear_*  
    NTN        model_zoo.py:331-346      NTN        -> k x txe_bilinear_pair_* + txe_linear_* (not reachable from model/model.py)
    BIM/LBM    model_zoo.py:301-328      BIM/LBM    -> txe_bilinear_pair_*  (+ score_all for the eval loop)
Graph argument: a taxoexpan_amd.graph.(Batched)DGLGraph -- the DGL-0.4 surface of the reference's loaders.
Side effects the callers rely on are kept: PGAT/PGCN pop g.ndata['pos'] (model_zoo.py:163,212); WeightedMeanReadout
writes g.ndata['a'] (:241).
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _fused_slope(activation):
    """slope of the activation if it is the leaky_relu the reference passes (model.py:25-41), else None"""
    if activation is F.leaky_relu:
        return ops.LEAKY_SLOPE
    if isinstance(activation, nn.LeakyReLU):
        return float(activation.negative_slope)
    return None


def _p(dropout_module_or_zero, training):
    """effective dropout probability of an nn.Dropout-or-falsy attribute"""
    if isinstance(dropout_module_or_zero, nn.Dropout) and training:
        return float(dropout_module_or_zero.p)
    return 0.0


# ---------------------------------------------------------------------------------------------------------------
# Graph propagation
# ---------------------------------------------------------------------------------------------------------------
def _maybe_dropout(p):
    """nn.Dropout for a truthy rate, else the falsy placeholder the reference keeps (0. / identity): only its rate is read here"""
    return nn.Dropout(p) if p else None


def _xavier_(*tensors, gain=1.414):
    for t in tensors:
        nn.init.xavier_normal_(t, gain=gain)


class GCNLayer(nn.Module):
    """parameters `weight` [in, out], `bias` [out] (model_zoo.py:14-32: both uniform in +-1/sqrt(out))"""

    def __init__(self, in_feats, out_feats, activation, dropout, bias=True):
        super().__init__()
        self.activation = activation
        self.dropout = _maybe_dropout(dropout) or 0.
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = nn.Parameter(torch.empty(out_feats)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        bound = self.weight.shape[1] ** -0.5
        with torch.no_grad():
            for t in (self.weight, self.bias):
                if t is not None:
                    t.uniform_(-bound, bound)

    def forward(self, g, h):
        """model_zoo.py:34-50 (the norm comes from in-degrees; g.ndata['norm'] is not needed)."""
        slope = _fused_slope(self.activation)
        cfg = ops.GCNConfig([self.weight.shape[1]], 0, [slope], [_p(self.dropout, self.training)], ops.new_seed())
        out = ops.apply_stack(ops.GCNStackFunction, g.csr(h.device), cfg, h, None, None, None, self.weight, self.bias, None)
        if self.activation and slope is None:
            out = self.activation(out)
        return out


class GATLayer(nn.Module):
    """parameters `fc.weight` [H*D, in], `attn_l` / `attn_r` [1, H, D], optional `res_fc.weight` (model_zoo.py:53-78: Xavier normal,
    gain 1.414)"""

    def __init__(self, in_dim, out_dim, num_heads=1, feat_drop=0.5, attn_drop=0.5, leaky_relu_alpha=0.2, residual=False):
        super().__init__()
        width = num_heads * out_dim
        self.num_heads, self.residual = num_heads, residual
        self.fc = nn.Linear(in_dim, width, bias=False)
        self.attn_l, self.attn_r = (nn.Parameter(torch.empty(1, num_heads, out_dim)) for _ in range(2))
        self.feat_drop, self.attn_drop = (_maybe_dropout(p) or (lambda x: x) for p in (feat_drop, attn_drop))
        self.leaky_relu = nn.LeakyReLU(leaky_relu_alpha)
        _xavier_(self.fc.weight.data, self.attn_l.data, self.attn_r.data)
        if residual:                                     # res_fc only when the widths differ, else the identity is added
            self.res_fc = nn.Linear(in_dim, width, bias=False) if in_dim != out_dim else None
            if self.res_fc is not None:
                _xavier_(self.res_fc.weight.data)

    @property
    def out_dim(self):
        return self.attn_l.shape[2]

    def forward(self, g, feature):
        """model_zoo.py:80-104 -> N x H x D'."""
        p_feat = _p(self.feat_drop, self.training)
        if self.residual and p_feat > 0:                # the residual reads the DROPPED input (model_zoo.py:82,100): drop it here
            feature, p_feat = F.dropout(feature, p_feat, True), 0.0
        cfg = ops.GATConfig([self.num_heads], [self.out_dim], [0], 0, self.leaky_relu.negative_slope, None,
                            p_feat, _p(self.attn_drop, self.training), "none", ops.new_seed())
        ret = ops.apply_stack(ops.GATStackFunction, g.csr(feature.device), cfg, feature, None, None, None, self.fc.weight, self.attn_l, self.attn_r, None)
        if self.residual:                               # model_zoo.py:98-103 (never enabled by model.py)
            if self.res_fc is not None:
                resval = ops.LinearFunction.apply(feature, None, self.res_fc.weight, None, 0).reshape((feature.shape[0], self.num_heads, -1))
            else:
                resval = torch.unsqueeze(feature, 1)
            ret = resval + ret
        return ret


def _gat_stack(layers, embeddings, g, h, pos, activation, training):
    """shared forward of GAT / PGAT: one fused autograd node when the activation is the reference's leaky_relu."""
    first = layers[0]
    slope = _fused_slope(activation)
    if slope is None or any(l.residual for l in layers):
        return None
    cfg = ops.GATConfig([l.num_heads for l in layers], [l.out_dim for l in layers],
                        [0 if embeddings is None else e.weight.shape[1] for e in (embeddings or layers)],
                        0 if embeddings is None else embeddings[0].weight.shape[0],
                        first.leaky_relu.negative_slope, slope, _p(first.feat_drop, training), _p(first.attn_drop, training),
                        "mean", ops.new_seed())
    params = []
    for i, l in enumerate(layers):
        params += [l.fc.weight, l.attn_l, l.attn_r, None if embeddings is None else embeddings[i].weight]
    if layers[-1].num_heads == 1 and not _NO_FOLD:
        # one-head output layer: a weighted-mean readout can fold it (ops 'collapse'); anything else materialises N x D
        return DeferredNodeOutput(g.csr(h.device), cfg, h, pos, params, ops.GATStackFunction)
    return ops.apply_stack(ops.GATStackFunction, g.csr(h.device), cfg, h, pos, None, None, *params)


# True switches the folded output layer off: graph_propagate then returns the N x out tensor (the parity tests compare the two routes;
# a plain module attribute like ops._NO_* -- nothing here reads the environment)
_NO_FOLD = False


class DeferredNodeOutput:
    """What PGAT / GAT (one-head output layer) and PGCN / GCN (activation-free output layer) .forward return: the N x out_dim
    node features, not yet computed.  MeanReadout / WeightedMeanReadout consume it through `.readout(...)` -- the output layer
    is then evaluated on G graph rows instead of N node rows (ops 'collapse'; same arithmetic, re-associated).  Every other
    use (attribute access, torch functions, other readouts) materialises the ordinary tensor once, with the same dropout seeds."""

    def __init__(self, csr, cfg, h, pos, params, fn):
        self._args = (csr, cfg, h, pos, params)
        self._fn = fn
        self._tensor = None

    def _out_dim(self):
        return self._args[1].out_dims[-1]

    def readout(self, rpos, pw):
        """the graph vectors [G, out_dim].  Without gradients: computed now ('collapse').  With gradients: a DeferredGraphVector --
        NOTHING is launched yet; the first consumer decides how the stack runs (a bilinear matcher on repeating queries takes the
        vector folded and hands the stack the query-side half of its work; anybody else gets the plain tensor)."""
        csr, cfg, h, pos, params = self._args
        if csr.n_nodes == 0 or csr.n_graphs == 0:       # empty batch: nothing to launch
            return h.new_zeros((csr.n_graphs, self._out_dim()), dtype=torch.float32)
        if torch.is_grad_enabled():
            return DeferredGraphVector(self, rpos, pw)
        return self._collapse("collapse", rpos, pw)[0]

    def _collapse(self, final, rpos, pw, fold_job=None):
        """run the stack with its output layer folded behind the readout: final = 'collapse' -> hg [G, D]; 'collapse_z' -> (Z [G, Kp], the
        output layer's packed weights Wp) with a fresh ops.FoldLink in the returned config"""
        import copy
        csr, cfg, h, pos, params = self._args
        c = copy.copy(cfg)
        c.final = final
        if final == "collapse_z":
            c.link, c.fold_job = ops.FoldLink(), fold_job
        return ops.apply_stack(self._fn, csr, c, h, pos, rpos, pw, *params), c

    def _can_fold(self):
        """may the stack stop at Z (the graph vector folded into the bilinear matcher)?"""
        csr, cfg = self._args[:2]
        if self._fn is ops.GCNStackFunction:
            return ops.gcn_folded_graph_vector_ok(csr, cfg, self._args[4])
        return self._fn is ops.GATStackFunction and ops.folded_graph_vector_ok(csr, cfg)

    def tensor(self):
        if self._tensor is None:
            csr, cfg, h, pos, params = self._args
            if csr.n_nodes == 0:
                self._tensor = h.new_zeros((0, self._out_dim()), dtype=torch.float32)
            else:
                self._tensor = ops.apply_stack(self._fn, csr, cfg, h, pos, None, None, *params)
        return self._tensor

    def __getattr__(self, name):                    # .shape, .device, .detach(), .cpu(), ... of the node features
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]

    def __len__(self):
        return len(self.tensor())

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        un = lambda a: a.tensor() if isinstance(a, (DeferredNodeOutput, DeferredGraphVector)) else a
        return func(*tree_map(un, tuple(args)), **tree_map(un, dict(kwargs or {})))


_SECOND_USE = ("taxoexpan_amd: this graph vector was consumed FOLDED by the bilinear matcher (hg = Z W^T never formed; "
               "ops.BilinearFoldedRunsFunction) and is now asked for as a tensor with gradients -- a second differentiable consumer is not "
               "supported after the fold.  Touch it before the matcher runs (hg.tensor()), or switch the fold off "
               "(taxoexpan_amd.ops._NO_MATCH_FOLD = True).  Under torch.no_grad() / via .detach() the values are available.")


class DeferredGraphVector:
    """What MeanReadout / WeightedMeanReadout return in grad mode: the graph vectors hg [G, D] of a propagation stack whose output layer
    folds behind the readout -- NOT YET COMPUTED.  The first consumer decides how the stack runs:
      * BIM / LBM on repeating query rows (`match_folded`): the stack stops at Z [G, Kp] ('collapse_z', hg = Z W^T is never formed) and
        the output layer's D x Kp product runs on one row per query run inside the matcher (ops.BilinearFoldedRunsFunction; same
        arithmetic, re-associated).  The matcher's query-side half is handed to the stack as a job, so it runs before the Z sweep.
      * anything else (`.tensor()`, any attribute / torch function / operator): the ordinary tensor, once ('collapse').
      * `.detach()` (logging hooks): the values without changing the route -- the stack runs as 'collapse_z' and hg = Z W^T is formed
        outside autograd.
    Needs no cooperation from the caller: the reference's model/model.py:70-87 drives it unchanged."""

    def __init__(self, node_out, rpos, pw):
        self._src = (node_out, rpos, pw)
        self._z = None              # (Z, Wp, FoldLink) once a 'collapse_z' stack has run
        self._tensor = None
        self._folded = False        # consumed by match_folded
        self._d = node_out._out_dim()

    @property
    def shape(self):
        return torch.Size((self._src[0]._args[0].n_graphs, self._d))

    @property
    def device(self):
        return self._src[0]._args[2].device

    def started(self):
        """has the propagation stack been launched for this vector?"""
        return self._z is not None or self._tensor is not None

    def can_fold(self):
        return (self._tensor is None and not self._folded and not ops._NO_MATCH_FOLD
                and (self._z is not None or self._src[0]._can_fold()))

    def _run_z(self, fold_job=None):
        if self._z is None:
            node, rpos, pw = self._src
            (Z, Wp), c = node._collapse("collapse_z", rpos, pw, fold_job)
            self._z = (Z, Wp, c.link)
        return self._z

    def match_folded(self, Wm, apply_exp, e2):
        """scores [G, 1] of the bilinear matcher (weight Wm [1, D, r]) against e2: the stacked query matrix [G, r] or an ops.RepeatedRows"""
        if not self.can_fold():
            raise RuntimeError("DeferredGraphVector.match_folded: the vector cannot be folded (any more)")
        stacked, rows, run_off = (None, e2.rows, e2.run_off) if isinstance(e2, ops.RepeatedRows) else (e2, None, None)
        job = ops.folded_match_job(stacked, rows, run_off, Wm) if self._z is None else None
        Z, Wp, link = self._run_z(job)
        self._folded = True
        return ops.BilinearFoldedRunsFunction.apply(Z, Wp, link, self._d, Wm, apply_exp, stacked, rows, run_off)

    def tensor(self):
        if self._tensor is None:
            if self._folded:
                if torch.is_grad_enabled():
                    raise RuntimeError(_SECOND_USE)
                return self.detach()
            if self._z is None:
                node, rpos, pw = self._src
                self._tensor = node._collapse("collapse", rpos, pw)[0]
            else:
                self._tensor = ops.FoldedGraphLinearFunction.apply(*self._z, self._d)
        return self._tensor

    def detach(self):
        """the values, outside autograd, WITHOUT deciding the route: a forward hook that logs `out.detach()` leaves the folded match in place"""
        if self._tensor is not None:
            return self._tensor.detach()
        if self._z is None and not self.can_fold():
            return self.tensor().detach()
        Z, Wp, link = self._run_z()
        return ops.folded_graph_linear(Z.detach(), Wp, self._d, link)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]

    def __len__(self):
        return int(self.shape[0])

    __torch_function__ = DeferredNodeOutput.__dict__["__torch_function__"]


def _delegate(name):
    def op(self, *args):
        return getattr(self.tensor(), name)(*[a.tensor() if isinstance(a, (DeferredNodeOutput, DeferredGraphVector)) else a for a in args])
    op.__name__ = name
    return op


for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__matmul__", "__rmatmul__",
           "__neg__", "__pow__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__iter__", "__repr__", "__bool__"):
    setattr(DeferredNodeOutput, _n, _delegate(_n))
    setattr(DeferredGraphVector, _n, _delegate(_n))
DeferredNodeOutput.__hash__ = object.__hash__
DeferredGraphVector.__hash__ = object.__hash__
DeferredGraphVector.__repr__ = lambda self: "DeferredGraphVector(shape=%s, %s)" % (tuple(self.shape), "folded" if self._folded else
                                                                                      ("tensor" if self._tensor is not None else
                                                                                       ("z" if self._z is not None else "pending")))


def _node_features(g):
    h = g.ndata['h']
    return h.tensor() if isinstance(h, DeferredNodeOutput) else h


def _gcn_layers(in_dim, hidden_dim, out_dim, extra, num_layers, activation, rates):
    """the num_layers + 1 GCNLayers of GCN / PGCN (model_zoo.py:117-126,140-153): in -> hidden -> ... -> hidden -> out, activation on
    all but the last, rates = (input, hidden, output) dropout; `extra` = width of the position embedding appended to every input"""
    widths = [in_dim] + [hidden_dim] * num_layers
    outs = [hidden_dim] * num_layers + [out_dim]
    acts = [activation] * num_layers + [None]
    drops = [rates[0]] + [rates[1]] * (num_layers - 1) + [rates[2]]
    return nn.ModuleList(GCNLayer(w + extra, o, a, r) for w, o, a, r in zip(widths, outs, acts, drops))


def _position_tables(n, vocab, dim):
    return nn.ModuleList(nn.Embedding(vocab, dim) for _ in range(n))


class GCN(nn.Module):
    def __init__(self, in_dim, hidden_dim, out_dim, num_layers, activation, in_dropout=0.1, hidden_dropout=0.1, output_dropout=0.0):
        super().__init__()
        self.layers = _gcn_layers(in_dim, hidden_dim, out_dim, 0, num_layers, activation, (in_dropout, hidden_dropout, output_dropout))

    def forward(self, g, features):
        """model_zoo.py:128-137"""
        out = _gcn_stack(self.layers, None, g, features, None, self.training)
        if out is not None:
            return out
        h = features                       # an activation other than leaky_relu: layer by layer (GCNLayer applies it itself)
        for layer in self.layers:
            h = layer(g, h)
        return h


class PGCN(nn.Module):
    def __init__(self, in_dim, hidden_dim, out_dim, pos_dim, num_layers, activation, in_dropout=0.1, hidden_dropout=0.1,
                 output_dropout=0.0, position_vocab_size=3):
        super().__init__()
        self.layers = _gcn_layers(in_dim, hidden_dim, out_dim, pos_dim, num_layers, activation,
                                  (in_dropout, hidden_dropout, output_dropout))
        self.prop_position_embeddings = _position_tables(num_layers + 1, position_vocab_size, pos_dim)

    def forward(self, g, features):
        """model_zoo.py:155-167 (pops g.ndata['pos'], :163)"""
        positions = g.ndata.pop('pos').to(features.device)
        out = _gcn_stack(self.layers, self.prop_position_embeddings, g, features, positions, self.training)
        if out is not None:
            return out
        h = features                       # an activation other than leaky_relu: layer by layer, concat materialised (model_zoo.py:164-166)
        for emb, layer in zip(self.prop_position_embeddings, self.layers):
            h = layer(g, torch.cat((h, emb(positions)), 1))
        return h


def _gcn_stack(layers, embeddings, g, h, pos, training):
    slopes = []
    for l in layers:
        s = _fused_slope(l.activation)
        if l.activation and s is None:
            return None                    # (the caller runs the layers one by one: same results, the activation applied by torch)
        slopes.append(s)
    vocab = 0 if embeddings is None else embeddings[0].weight.shape[0]
    cfg = ops.GCNConfig([l.weight.shape[1] for l in layers], vocab, slopes, [_p(l.dropout, training) for l in layers],
                        ops.new_seed())
    params = []
    for i, l in enumerate(layers):
        params += [l.weight, l.bias, None if embeddings is None else embeddings[i].weight]
    if slopes[-1] is None and not _NO_FOLD:
        # activation-free output layer: a weighted-mean readout can fold it (ops 'collapse'); anything else materialises N x out
        return DeferredNodeOutput(g.csr(h.device), cfg, h, pos, params, ops.GCNStackFunction)
    return ops.apply_stack(ops.GCNStackFunction, g.csr(h.device), cfg, h, pos, None, None, *params)


def _gat_layers(in_dim, hidden_dim, out_dim, extra, num_layers, heads, feat_drop, attn_drop, alpha, residual):
    """the num_layers + 1 GATLayers of GAT / PGAT (model_zoo.py:170-181,193-208): hidden layers concatenate their heads, so layer l
    reads hidden_dim * heads[l-1] (+ extra) columns; the output layer has heads[-1] heads; the first layer never has a residual"""
    ins = [in_dim] + [hidden_dim * heads[l - 1] for l in range(1, num_layers)] + [hidden_dim * heads[-2]]
    outs = [hidden_dim] * num_layers + [out_dim]
    hs = list(heads[:num_layers]) + [heads[-1]]
    res = [False] + [residual] * num_layers
    return nn.ModuleList(GATLayer(i + extra, o, h, feat_drop, attn_drop, alpha, r) for i, o, h, r in zip(ins, outs, hs, res))


class GAT(nn.Module):
    def __init__(self, in_dim, hidden_dim, out_dim, num_layers, heads, activation, feat_drop=0.5, attn_drop=0.5,
                 leaky_relu_alpha=0.2, residual=False):
        super().__init__()
        self.num_layers, self.activation = num_layers, activation
        self.gat_layers = _gat_layers(in_dim, hidden_dim, out_dim, 0, num_layers, heads, feat_drop, attn_drop, leaky_relu_alpha, residual)

    def forward(self, g, features):
        """model_zoo.py:183-190"""
        out = _gat_stack(self.gat_layers, None, g, features, None, self.activation, self.training)
        if out is not None:
            return out
        h = features
        for l in range(self.num_layers):
            h = self.gat_layers[l](g, h).flatten(1)
            h = self.activation(h)
        return self.gat_layers[-1](g, h).mean(1)


class PGAT(nn.Module):
    def __init__(self, in_dim, hidden_dim, out_dim, pos_dim, num_layers, heads, activation, feat_drop=0.5, attn_drop=0.5,
                 leaky_relu_alpha=0.2, residual=False, position_vocab_size=3):
        super().__init__()
        self.num_layers, self.activation = num_layers, activation
        self.gat_layers = _gat_layers(in_dim, hidden_dim, out_dim, pos_dim, num_layers, heads, feat_drop, attn_drop, leaky_relu_alpha,
                                      residual)
        self.prop_position_embeddings = _position_tables(num_layers + 1, position_vocab_size, pos_dim)

    def forward(self, g, features):
        """model_zoo.py:210-220 (pops g.ndata['pos'], :212)"""
        positions = g.ndata.pop('pos').to(features.device)
        out = _gat_stack(self.gat_layers, self.prop_position_embeddings, g, features, positions, self.activation, self.training)
        if out is not None:
            return out
        h = features                       # generic activation / residual: layer by layer, concat materialised
        for l in range(self.num_layers):
            p = self.prop_position_embeddings[l](positions)
            h = self.gat_layers[l](g, torch.cat((h, p), 1)).flatten(1)
            h = self.activation(h)
        p = self.prop_position_embeddings[-1](positions)
        return self.gat_layers[-1](g, torch.cat((h, p), 1)).mean(1)


# ---------------------------------------------------------------------------------------------------------------
# Readouts
# ---------------------------------------------------------------------------------------------------------------
class MeanReadout(nn.Module):
    def __init__(self):
        super(MeanReadout, self).__init__()

    def forward(self, g, pos=None):
        """model_zoo.py:231-232"""
        h = g.ndata['h']
        if isinstance(h, DeferredNodeOutput):
            return h.readout(None, None)
        return ops.ReadoutFunction.apply(g.csr(h.device), h, None, None)


class WeightedMeanReadout(nn.Module):
    def __init__(self):
        super(WeightedMeanReadout, self).__init__()
        self.position_weights = nn.Embedding(3, 1)
        self.nonlinear = F.softplus

    def forward(self, g, pos):
        """model_zoo.py:240-242"""
        h = g.ndata['h']
        g.ndata['a'] = _LazyPositionWeight(self.position_weights.weight, pos)
        if isinstance(h, DeferredNodeOutput):
            return h.readout(pos, self.position_weights.weight)
        return ops.ReadoutFunction.apply(g.csr(h.device), h, pos, self.position_weights.weight)


class ConcatReadout(nn.Module):
    def __init__(self):
        super(ConcatReadout, self).__init__()

    def forward(self, g, pos):
        """model_zoo.py:248-258: [sum_{pos=0} h / n, mean_{pos=1} h, sum_{pos=2} h / n]"""
        h = _node_features(g)
        return ops.ReadoutMultiFunction.apply(g.csr(h.device), h, pos, 3)


class SumReadout(nn.Module):
    def __init__(self):
        super(SumReadout, self).__init__()

    def forward(self, g):
        """model_zoo.py:265-267"""
        h = _node_features(g)
        return ops.ReadoutMultiFunction.apply(g.csr(h.device), h, None, 1)


class MaxReadout(nn.Module):
    def __init__(self):
        super(MaxReadout, self).__init__()

    def forward(self, g):
        """model_zoo.py:274-276"""
        h = _node_features(g)
        return ops.ReadoutMultiFunction.apply(g.csr(h.device), h, None, 2)


class _LazyPositionWeight:
    """g.ndata['a'] of model_zoo.py:241 (softplus(Emb[pos])) -- nobody on the hot path reads it, so it is only
    materialised if a caller asks (`.tensor()`)."""

    def __init__(self, weight, pos):
        self._w, self._pos = weight, pos

    def tensor(self):
        return F.softplus(self._w.detach()[self._pos.long()])


# ---------------------------------------------------------------------------------------------------------------
# Matchers
# ---------------------------------------------------------------------------------------------------------------
def _graph_vector(e1):
    """the plain [G, D] tensor of a matcher's first argument (a readout's DeferredGraphVector: materialised now)"""
    return e1.tensor() if isinstance(e1, (DeferredGraphVector, DeferredNodeOutput)) else e1


class MLP(nn.Module):
    def __init__(self, l_dim, r_dim, hidden_dim):
        super(MLP, self).__init__()
        activation = nn.ReLU()
        self.ffn = nn.Sequential(          # parameter container: same names / shapes / init as the reference (model_zoo.py:285-289)
            nn.Linear(l_dim + r_dim, hidden_dim),
            activation,
            nn.Linear(hidden_dim, 1)
        )

    def forward(self, e1, e2):
        """model_zoo.py:291-298: ffn(cat(e1, e2)); the concat is synthesised by the GEMM's operand loader"""
        e1, e2 = _graph_vector(e1), ops.dense_rows(e2)
        if e1.shape[0] == 0:
            return e1.new_zeros((0, 1), dtype=torch.float32)
        hid = ops.LinearFunction.apply(e1, e2, self.ffn[0].weight, self.ffn[0].bias, 1)
        return ops.LinearFunction.apply(hid, None, self.ffn[2].weight, self.ffn[2].bias, 0)


class _Bilinear(nn.Module):
    apply_exp = False

    def __init__(self, l_dim, r_dim):
        super(_Bilinear, self).__init__()
        self.W = nn.Bilinear(l_dim, r_dim, 1, bias=False)      # parameter container: same name/shape/init as the reference

    def forward(self, e1, e2):
        """e1 (*, l_dim), e2 (*, r_dim) -> (*, 1).  ONE route per call, decided by _route and recorded (ops.ROUTES['match'])."""
        if e1.shape[0] == 0:                               # empty batch
            return e2.new_zeros((0, 1), dtype=torch.float32) if torch.is_tensor(e2) else e2.rows.new_zeros((0, 1), dtype=torch.float32)
        route, e2 = self._route(e1, e2)
        ops.note_route("match", route)
        W, ex = self.W.weight, self.apply_exp
        if route == "folded":                              # hg = Z W^T never formed: the output layer's product runs on the query runs
            return e1.match_folded(W, ex, e2)
        if route == "expand":                              # the eval loop's `nf.expand(n_position, -1)` (test_fast.py:122-123)
            return ops.score_block(e2[:1], ops.bilinear_project(_graph_vector(e1), W), ex).reshape(-1, 1)
        # pair form: V = e2 W^T needs neither the graph nor the encoder -- with the encoder not launched yet it goes to the second stream
        lazy = isinstance(e1, DeferredGraphVector) and not e1.started()
        pre = ops.bilinear_query_prefetch(e2, W) if (route == "pair" and lazy and torch.is_grad_enabled()) else None
        hg = _graph_vector(e1)
        if route == "runs":
            return ops.BilinearRunsFunction.apply(hg, W, ex, e2.rows, e2.run_off)
        if route == "stacked":
            return ops.BilinearStackedRunsFunction.apply(hg, e2, W, ex)
        return ops.BilinearPairFunction.apply(hg, e2, W, ex, pre)

    def _route(self, e1, e2):
        """(route, e2): 'folded' | 'runs' (ops.RepeatedRows) | 'stacked' (repeating rows of a stacked matrix, found on the device) |
        'expand' (one query against all candidates, no grad) | 'pair' (one GEMM row per pair).  e2 comes back dense where the route
        needs it so."""
        n, grad = e1.shape[0], torch.is_grad_enabled()
        if isinstance(e2, ops.RepeatedRows):
            if e2.requires_grad or e2.n_rows != n or 4 * e2.rows.shape[0] > e2.n_rows:     # (hardly any repetition: the GEMM form)
                return "pair", e2.dense()
            runs = "runs"
        elif e2.dim() == 2 and e2.stride(0) == 0 and e2.shape[0] == n and not grad:
            return "expand", e2
        else:
            runs = "stacked" if (self._stacked_runs_ok(n, e2) and self._repeats(e2)) else None
        if runs is not None and grad and isinstance(e1, DeferredGraphVector) and e1.can_fold():
            return "folded", e2
        return (runs or "pair"), e2

    # ---- stacked query rows that repeat (data_loaders.py:9-28 stacks a query's row once per pair) -------------------------------------
    def _stacked_runs_ok(self, n, e2):
        return (not ops._NO_QUERY_RUNS and torch.is_grad_enabled() and torch.is_tensor(e2) and e2.is_cuda and e2.dim() == 2 and
                not e2.requires_grad and 256 <= e2.shape[0] <= (1 << 18) and e2.shape[0] == n)     # (one workgroup scans the rows)

    RECHECK_EVERY = 64      # training batches between two looks at the run count

    def _repeats(self, e2):
        """does this matcher's training input repeat its query rows?  At least three rows in four repeat -> the one-row-per-run form
        (which finds its runs on the device in every call, so a batch that repeats less is only slower, never wrong), otherwise the
        GEMM form.  Decided on the first training batch (the runs are counted on the device and read back: the one host
        synchronisation of the scheme) and RE-decided every RECHECK_EVERY batches: the count of batch k * RECHECK_EVERY goes to pinned
        memory asynchronously and is applied exactly RECHECK_EVERY / 2 calls later (the copy finished long before; the wait is a formality)
        -- at a call count, not at whatever moment the copy happened to land, so a run with fixed seeds takes the same route at the same
        step every time, on every rank.  A loader that starts with one odd batch, or changes its collate, is followed within
        1.5 RECHECK_EVERY batches."""
        st = self.__dict__.get("_runs_watch")
        if st is None:
            n_runs = int(ops.find_row_runs(e2)[2].item())
            st = self.__dict__["_runs_watch"] = dict(dec=bool(4 * n_runs <= e2.shape[0]), calls=0, pending=None)
            return st["dec"]
        st["calls"] += 1
        if st["pending"] is not None and st["calls"] >= st["pending"][3]:
            ev, buf, G, _due = st["pending"]
            ev.synchronize()
            st["dec"], st["pending"] = bool(4 * int(buf[0]) <= G), None
        if st["calls"] % self.RECHECK_EVERY == 0 and st["pending"] is None:
            buf = st.get("buf")
            if buf is None:
                buf = st["buf"] = torch.empty(1, dtype=torch.int32).pin_memory()
            buf.copy_(ops.find_row_runs(e2)[2], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            st["pending"] = (ev, buf, int(e2.shape[0]), st["calls"] + self.RECHECK_EVERY // 2)
        return st["dec"]

    def score_all(self, hg, queries, block=None, out=None):
        """The whole scoring loop at once: S[q][g] = match(hg[g], queries[q]) (test_fast.py:116-123)."""
        from .scoring import score_all
        return score_all(self, hg, queries, block=block, out=out)


class BIM(_Bilinear):
    """model_zoo.py:301-313"""
    apply_exp = False


class LBM(_Bilinear):
    """model_zoo.py:316-328: exp of the bilinear form"""
    apply_exp = True


class NTN(nn.Module):
    def __init__(self, l_dim, r_dim, k=4, non_linear=torch.tanh):
        super(NTN, self).__init__()
        self.u_R = nn.Linear(k, 1, bias=False)                  # parameter containers: names / shapes / init of model_zoo.py:332-337
        self.f = non_linear
        self.W = nn.Bilinear(l_dim, r_dim, k, bias=True)
        self.V = nn.Linear(l_dim + r_dim, k, bias=False)

    def forward(self, e1, e2):
        """model_zoo.py:339-346: u_R(f(W(e1, e2) + V(cat(e1, e2)))) -> (*, 1); one bilinear slice per output, the concat is virtual"""
        e1, e2 = _graph_vector(e1), ops.dense_rows(e2)
        k = self.W.weight.shape[0]
        bil = torch.cat([ops.BilinearPairFunction.apply(e1, e2, self.W.weight[j:j + 1], False).reshape(-1, 1) for j in range(k)], 1)
        lin = ops.LinearFunction.apply(e1, e2, self.V.weight, None, 0)
        return ops.LinearFunction.apply(self.f(bil + self.W.bias + lin), None, self.u_R.weight, None, 0)

