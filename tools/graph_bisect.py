#!/usr/bin/env python3
"""which part of the training step survives HIP-graph capture?  python tools/graph_bisect.py <stage> [workload]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import synthetic as syn  # noqa: E402
from taxoexpan_amd.loss import info_nce_loss  # noqa: E402
from taxoexpan_amd.optim import Adam  # noqa: E402

from taxoexpan_amd import _lib, ops  # noqa: E402
_skip = [t for t in os.environ.get("TXE_SKIP", "").split(",") if t]
_seen = []
_orig = _lib.call


def _call(name, *args):
    if name not in _seen:
        _seen.append(name)
    if any(name == t for t in _skip):
        return 0
    return _orig(name, *args)


_lib.call = ops.call = _call
stage = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "pgcn"
dev = torch.device("cuda:0")
torch.manual_seed(47)
tax = syn.make_named_taxonomy("mag_cs", seed=47)
model = bench.make_model(workload, dev)
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
b = bench.build_batches(tax, 1, 1000, dev)[0]
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)


def body():
    g = b["g"]
    g.ndata["pos"] = b["pos"]
    if stage == "score":
        from taxoexpan_amd import ops
        U = ops.bilinear_project(b["qf"][:, :250].new_zeros((1000, 500)).normal_(), model.match.W.weight)
        return ops.score_block(b["qf"][:256], U, True)
    if stage == "fwd_eval":
        with torch.no_grad():
            return model(g, b["x"], b["qf"])
    if stage == "zero":
        opt.zero_grad(set_to_none=True)
        return None
    pred = model(g, b["x"], b["qf"])
    if stage == "fwd":
        return pred
    if stage == "fwd_sum_bwd":
        opt.zero_grad(set_to_none=True)
        pred.sum().backward()
        return None
    loss = info_nce_loss(pred.reshape(bench.N_QUERIES, -1), target)
    if stage == "loss":
        return loss
    opt.zero_grad(set_to_none=True)
    loss.backward()
    if stage == "bwd":
        return None
    opt.step()


for _ in range(3):
    body()
if stage == "fwd_eval":
    model.eval()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("CAPTURE OK", stage, workload, flush=True)
print("CALLS", ",".join(_seen))
