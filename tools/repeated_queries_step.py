#!/usr/bin/env python3
"""The BASELINE training step with the query features stacked per pair (the reference's collate) against ops.RepeatedRows (one row per
query): resident batches, 50 steps each, and the kernels that differ.    python tools/repeated_queries_step.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import ops, synthetic as syn  # noqa: E402
from taxoexpan_amd.optim import Adam  # noqa: E402

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = bench.make_model("pgat", dev)
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
batches = bench.build_batches(tax, 4, 1000, dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
runs = []
for b in batches:
    q = b["qf"].cpu().numpy()
    qid = np.concatenate([[0], np.cumsum(np.any(q[1:] != q[:-1], axis=1))])
    first = np.concatenate([[True], qid[1:] != qid[:-1]])
    runs.append(dict(b, qf=ops.RepeatedRows.from_ids(torch.from_numpy(q[first]).to(dev), qid)))
    assert runs[-1]["qf"].rows.shape[0] == bench.N_QUERIES


def timed(bs, n=50):
    for i in range(10):
        bench.train_step(model, opt, bs[i % 4], target, 1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        bench.train_step(model, opt, bs[i % 4], target, 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for rep in range(2):
    print(f"stacked rows {timed(batches):.4f} ms/step   repeated rows {timed(runs):.4f} ms/step", flush=True)
for name, bs in (("stacked", batches), ("repeated", runs)):
    acc = {}
    for i in range(8):
        for k, sec, work, kind in bench.profile_step(model, opt, bs[i % 4], target):
            a = acc.setdefault(k, [0.0, 0])
            a[0] += sec
            a[1] += 1
    print(name, {k: round(v[0] / 8 * 1e6, 1) for k, v in acc.items() if any(p in k for p in ("gemm_kernel<true, true, 2", "gemm_kernel<false, false, 4, 2", "gat_aggregate_fwd", "runs_", "rowdot", "bil_scale", "reduce_splits"))})
