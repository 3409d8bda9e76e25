"""Model assembly behind the reference's interface (model/model.py:13-87): `TaxoExpan(propagation_method, readout_method,
matching_method, **options)`, `.forward(g, h, qf)`, and the three sub-modules reachable as `.graph_propagate`, `.readout`,
`.match` (test_fast.py:25-28 and infer.py:15-18 reach into them directly).  The method strings select constructors from
tables; a string that is in no table simply leaves the attribute unset -- the reference's `assert "<message>"` on that branch
(model/model.py:43,58,67) never fires either, so scripts behave the same."""
import torch.nn
import torch.nn.functional

from . import model_zoo as zoo

_LEAKY = torch.nn.functional.leaky_relu        # the activation model/model.py hands to every propagation module


def _gcn_like(cls, positional):
    def make(o):
        dims = (o["in_dim"], o["hidden_dim"], o["out_dim"]) + ((o["pos_dim"],) if positional else ())
        return cls(*dims, num_layers=o["num_layers"], activation=_LEAKY, in_dropout=o["feat_drop"],
                   hidden_dropout=o["hidden_drop"], output_dropout=o["out_drop"])
    return make


def _gat_like(cls, positional):
    def make(o):
        dims = (o["in_dim"], o["hidden_dim"], o["out_dim"]) + ((o["pos_dim"],) if positional else ())
        return cls(*dims, num_layers=o["num_layers"], heads=o["heads"], activation=_LEAKY, feat_drop=o["feat_drop"],
                   attn_drop=o["attn_drop"])
    return make


PROPAGATION = {"GCN": _gcn_like(zoo.GCN, False), "PGCN": _gcn_like(zoo.PGCN, True),
               "GAT": _gat_like(zoo.GAT, False), "PGAT": _gat_like(zoo.PGAT, True)}
# readout -> (module class, how many copies of out_dim the graph vector holds)
READOUT = {"MR": (zoo.MeanReadout, 1), "WMR": (zoo.WeightedMeanReadout, 1), "CR": (zoo.ConcatReadout, 3)}
MATCH = {"MLP": lambda l, r, o: zoo.MLP(l, r, o["hidden_dim"]), "LBM": lambda l, r, o: zoo.LBM(l, r),
         "BIM": lambda l, r, o: zoo.BIM(l, r)}


class TaxoExpan(torch.nn.Module):
    def __init__(self, propagation_method, readout_method, matching_method, **options):
        super().__init__()
        self.propagation_method, self.readout_method, self.matching_method = propagation_method, readout_method, matching_method
        self.options = options
        if propagation_method in PROPAGATION:
            self.graph_propagate = PROPAGATION[propagation_method](options)
        if readout_method in READOUT:
            cls, copies = READOUT[readout_method]
            self.readout = cls()
            dims = (options["out_dim"] * copies, options["in_dim"])        # (graph side, query side) of the matcher
            if matching_method in MATCH:
                self.match = MATCH[matching_method](dims[0], dims[1], options)

    def forward(self, g, h, qf):
        """model/model.py:70-87, statement for statement: positions are read BEFORE propagation (PGAT / PGCN pop them), node states are
        left in g.ndata['h'], one score per (egonet, query) row comes back.  Nothing here knows about the routes below it: in grad mode
        graph_propagate and readout only DESCRIBE their work (model_zoo.DeferredNodeOutput / DeferredGraphVector) and the matcher, the
        first one to hold both the graph vector and the queries, decides how the stack runs -- so the reference's own model/model.py with
        its import swapped takes exactly the same route."""
        pos = g.ndata['pos'].to(h.device)
        g.ndata['h'] = self.graph_propagate(g, h)
        hg = self.readout(g, pos)
        return self.match(hg, qf)

    def __str__(self):
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return super().__str__() + '\nTrainable parameters: {}'.format(n)


def encode_graph(model, bg, h, pos):
    """test_fast.py:25-28 / infer.py:15-18: candidate egonets -> one vector per egonet"""
    bg.ndata['h'] = model.graph_propagate(bg, h)
    return model.readout(bg, pos)
