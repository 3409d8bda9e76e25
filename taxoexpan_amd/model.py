"""Model assembly with the reference's interface (model/model.py:13-87): TaxoExpan(propagation_method,
readout_method, matching_method, **options), .forward(g, h, qf), attributes .graph_propagate / .readout / .match
(test_fast.py:25-28 and infer.py:15-18 reach into those directly).  Unknown method strings fall through silently,
like the reference's `assert "<string>"` (model/model.py:43,58,67)."""
import torch.nn as nn
import torch.nn.functional as F

from .model_zoo import BIM, GAT, GCN, LBM, MLP, PGAT, PGCN, ConcatReadout, MeanReadout, WeightedMeanReadout


class TaxoExpan(nn.Module):
    def __init__(self, propagation_method, readout_method, matching_method, **options):
        super(TaxoExpan, self).__init__()
        self.propagation_method = propagation_method
        self.readout_method = readout_method
        self.matching_method = matching_method
        self.options = options
        o = options
        if propagation_method == "GCN":
            self.graph_propagate = GCN(o["in_dim"], o["hidden_dim"], o["out_dim"], num_layers=o["num_layers"],
                                       activation=F.leaky_relu, in_dropout=o["feat_drop"], hidden_dropout=o["hidden_drop"],
                                       output_dropout=o["out_drop"])
        elif propagation_method == "PGCN":
            self.graph_propagate = PGCN(o["in_dim"], o["hidden_dim"], o["out_dim"], o["pos_dim"], num_layers=o["num_layers"],
                                        activation=F.leaky_relu, in_dropout=o["feat_drop"], hidden_dropout=o["hidden_drop"],
                                        output_dropout=o["out_drop"])
        elif propagation_method == "GAT":
            self.graph_propagate = GAT(o["in_dim"], o["hidden_dim"], o["out_dim"], num_layers=o["num_layers"], heads=o["heads"],
                                       activation=F.leaky_relu, feat_drop=o["feat_drop"], attn_drop=o["attn_drop"])
        elif propagation_method == "PGAT":
            self.graph_propagate = PGAT(o["in_dim"], o["hidden_dim"], o["out_dim"], o["pos_dim"], num_layers=o["num_layers"],
                                        heads=o["heads"], activation=F.leaky_relu, feat_drop=o["feat_drop"],
                                        attn_drop=o["attn_drop"])

        if readout_method == "MR":
            self.readout = MeanReadout()
            l_dim, r_dim = o["out_dim"], o["in_dim"]
        elif readout_method == "WMR":
            self.readout = WeightedMeanReadout()
            l_dim, r_dim = o["out_dim"], o["in_dim"]
        elif readout_method == "CR":
            self.readout = ConcatReadout()
            l_dim, r_dim = o["out_dim"] * 3, o["in_dim"]

        if matching_method == "MLP":
            self.match = MLP(l_dim, r_dim, o["hidden_dim"])
        elif matching_method == "LBM":
            self.match = LBM(l_dim, r_dim)
        elif matching_method == "BIM":
            self.match = BIM(l_dim, r_dim)

    def forward(self, g, h, qf):
        """model/model.py:70-87"""
        pos = g.ndata['pos'].to(h.device)
        g.ndata['h'] = self.graph_propagate(g, h)
        hg = self.readout(g, pos)
        scores = self.match(hg, qf)
        return scores

    def __str__(self):
        n = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return super(TaxoExpan, self).__str__() + '\nTrainable parameters: {}'.format(n)


def encode_graph(model, bg, h, pos):
    """test_fast.py:25-28 / infer.py:15-18"""
    bg.ndata['h'] = model.graph_propagate(bg, h)
    hg = model.readout(bg, pos)
    return hg
