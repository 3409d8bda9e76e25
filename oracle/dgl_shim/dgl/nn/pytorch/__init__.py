"""TEST INFRASTRUCTURE ONLY -- dgl.nn.pytorch.edge_softmax (model_zoo.py:6,112).

[DGL 0.4 published semantics, parity-unpinned] for every destination node v and
every trailing index, softmax over the logits of the edges that END in v:
    a_e = exp(s_e - max_{e' -> v} s_e') / sum_{e' -> v} exp(s_e' - max)
"""
import torch

from . import glob  # noqa: F401


def edge_softmax(g, logits):
    dst = g._dst.to(logits.device)
    n = g._n
    tail = tuple(logits.shape[1:])
    idx = dst.reshape((-1,) + (1,) * len(tail)).expand_as(logits)
    mx = torch.full((n,) + tail, float("-inf"), dtype=logits.dtype, device=logits.device)
    mx = mx.scatter_reduce(0, idx, logits, reduce="amax", include_self=True)
    ex = torch.exp(logits - mx[dst])
    den = torch.zeros((n,) + tail, dtype=logits.dtype, device=logits.device).index_add(0, dst, ex)
    return ex / den[dst]
