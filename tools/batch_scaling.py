#!/usr/bin/env python3
"""How do the step's kernels scale with the batch?  The training step on 1x, 2x, 4x, 8x the BASELINE batch (4,096 egonets): per-kernel
average duration per 4,096 egonets.  A kernel whose figure falls with the batch pays a fixed cost per launch (ramp-up, drain, the
dependent-load chain of its first wave) that a bigger batch amortises; one whose figure stays is throughput-bound.
    python tools/batch_scaling.py [pattern ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import synthetic as syn  # noqa: E402
from taxoexpan_amd.optim import Adam  # noqa: E402

pats = sys.argv[1:]
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
tax = syn.make_named_taxonomy("mag_cs", seed=47)
rows = {}
BASE = bench.N_QUERIES
for k in (1, 2, 4, 8):
    bench.N_QUERIES = BASE * k
    torch.manual_seed(47)
    model = bench.make_model("pgat", dev)
    opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
    batches = bench.build_batches(tax, 2, 1000, dev)
    target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
    for i in range(6):
        bench.train_step(model, opt, batches[i % 2], target, 1)
    torch.cuda.synchronize()
    acc = {}
    n = 6
    for i in range(n):
        for name, sec, work, kind in bench.profile_step(model, opt, batches[i % 2], target):
            a = acc.setdefault(name, [0.0, 0])
            a[0] += sec
            a[1] += 1
    for name, (sec, cnt) in acc.items():
        rows.setdefault(name, {})[k] = sec / n * 1e6 / k          # us per step per 4,096 egonets
    print(k, "x:", sum(batches[i]["n_nodes"] for i in range(2)) / 2, "nodes", flush=True)
print(f"{'kernel':60s} {'1x':>8s} {'2x':>8s} {'4x':>8s} {'8x':>8s}   us per 4,096 egonets")
for name, r in sorted(rows.items(), key=lambda kv: -kv[1].get(1, 0)):
    if pats and not any(p in name for p in pats):
        continue
    print(f"{name[:60]:60s} {r.get(1, 0):8.1f} {r.get(2, 0):8.1f} {r.get(4, 0):8.1f} {r.get(8, 0):8.1f}")
