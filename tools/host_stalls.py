#!/usr/bin/env python3
"""Per-step host time of 400 training steps with Python's cyclic collector as it comes, and after gc.collect() + gc.freeze():
where do multi-millisecond host stalls come from (a ~100 ms generation-2 pass every few hundred steps; ~12 ms waits are the launch
queue's back-pressure when the host runs ahead of the GPU).   gpurun -- python tools/host_stalls.py"""
import os, sys, time, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from taxoexpan_amd import TaxoExpan, synthetic as syn
from taxoexpan_amd.optim import Adam
dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = TaxoExpan("PGAT", "WMR", "LBM", **bench.MAG).to(dev).train()
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
batches = bench.build_batches(tax, 4, 1000, dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
for mode in ("gc on", "gc off", "gc off"):
    if mode == "gc off":
        gc.collect(); gc.freeze()
    for i in range(20):
        bench.train_step(model, opt, batches[i % 4], target, 1)
    torch.cuda.synchronize()
    ts = []
    for i in range(400):
        t0 = time.perf_counter()
        bench.train_step(model, opt, batches[i % 4], target, 1)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    big = [(i, round(t * 1e3, 2)) for i, t in enumerate(ts) if t > 3e-3]
    print(mode, "host ms median", round(sorted(ts)[200] * 1e3, 3), "stalls >3ms:", big[:12])
