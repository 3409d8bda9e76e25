#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one training step (no GPU wait)?  Compared with the GPU-side step time this
tells whether the step is launch-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import TaxoExpan, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
variant = (os.environ.get("TXE_VARIANT", "PGAT WMR LBM")).split()
model = TaxoExpan(*variant, **bench.MAG).to(dev).train()
from taxoexpan_amd.optim import Adam  # noqa: E402
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
batches = bench.build_batches(tax, 2, 1000, dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
for i in range(5):
    bench.train_step(model, opt, batches[i % 2], target, 1)
torch.cuda.synchronize()
host = []
for i in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.train_step(model, opt, batches[i % 2], target, 1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0, t2 - t0))
print("host enqueue ms:", [round(h[0] * 1e3, 2) for h in host])
print("step wall ms   :", [round(h[1] * 1e3, 2) for h in host])
# break down the host part
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    bench.train_step(model, opt, batches[i % 2], target, 1)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
