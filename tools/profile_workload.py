#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes: a known-size device copy (calibrates FETCH_SIZE / WRITE_SIZE, which on
gfx950 under-report wide streaming reads by 2x -- MI355X_MICROARCH.md HBM section) followed by training steps of the
bench workload (same model / batches as bench.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import TaxoExpan, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = TaxoExpan("PGAT", "WMR", "LBM", **bench.MAG).to(dev).train()
from taxoexpan_amd.optim import Adam  # noqa: E402
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
batches = bench.build_batches(tax, 2, 1000, dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
for i in range(3):
    bench.train_step(model, opt, batches[i % 2], target, 1)
torch.cuda.synchronize()
# calibration: 1 GiB read + 1 GiB write by one elementwise copy kernel (far larger than the 256 MiB Infinity Cache)
src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
torch.cuda.synchronize()
dst.copy_(src)
torch.cuda.synchronize()
print("CALIB_BYTES", src.numel() * 4)
for i in range(int(os.environ.get("TXE_PROF_STEPS", "4"))):
    bench.train_step(model, opt, batches[i % 2], target, 1)
torch.cuda.synchronize()
print("N", [b["n_nodes"] for b in batches], "E", [b["n_edges"] for b in batches])
