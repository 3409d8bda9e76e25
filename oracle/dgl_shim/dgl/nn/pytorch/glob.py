"""TEST INFRASTRUCTURE ONLY -- dgl.nn.pytorch.glob.SumPooling / MaxPooling
(model_zoo.py:7,263,272): per-graph sum / max of a node feature tensor."""
import torch
import torch.nn as nn


class SumPooling(nn.Module):
    def forward(self, graph, feat):
        import dgl
        return dgl._seg_sum(graph, feat)


class MaxPooling(nn.Module):
    def forward(self, graph, feat):
        outs, off = [], 0
        for n in graph.batch_num_nodes:
            outs.append(feat[off:off + n].max(0)[0])
            off += n
        return torch.stack(outs)
