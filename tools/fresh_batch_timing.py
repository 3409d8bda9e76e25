#!/usr/bin/env python3
"""where does a step with a freshly built batch spend its time?  host time of the build (side stream) and of the step's enqueue,
against the loop's wall time per step"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import graph as Gr, synthetic as syn  # noqa: E402
from taxoexpan_amd.optim import Adam  # noqa: E402

dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = bench.make_model("pgat", dev)
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
dtax = Gr.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
torch.cuda.synchronize()
for mode in ("inline", "side"):
    side = torch.cuda.Stream(device=dev) if mode == "side" else None
    for i in range(5):
        bench.train_step(model, opt, bench.fresh_batch(tax, dtax, 100 + i, dev, side), target, 1)
    torch.cuda.synchronize()
    tb = ts = 0.0
    t00 = time.perf_counter()
    n = 30
    for i in range(n):
        t0 = time.perf_counter()
        b = bench.fresh_batch(tax, dtax, 200 + i, dev, side)
        t1 = time.perf_counter()
        bench.train_step(model, opt, b, target, 1)
        t2 = time.perf_counter()
        tb += t1 - t0
        ts += t2 - t1
    torch.cuda.synchronize()
    print(f"{mode}: build host {tb / n * 1e3:.3f} ms, step enqueue {ts / n * 1e3:.3f} ms, loop wall {(time.perf_counter() - t00) / n * 1e3:.3f} ms/step")
import cProfile, pstats
side = torch.cuda.Stream(device=dev)
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    b = bench.fresh_batch(tax, dtax, 300 + i, dev, side)
    bench.train_step(model, opt, b, target, 1)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
pstats.Stats(pr).sort_stats("cumtime").print_stats(70)

# un-instrumented segment clocks of one iteration (host time only)
from taxoexpan_amd.loss import info_nce_loss
seg = dict(build=0.0, zero=0.0, fwd=0.0, loss=0.0, bwd=0.0, opt=0.0)
n = 30
for i in range(n):
    t0 = time.perf_counter(); b = bench.fresh_batch(tax, dtax, 400 + i, dev, side)
    t1 = time.perf_counter(); g = b["g"]; g.ndata["pos"] = b["pos"]; opt.zero_grad(set_to_none=True)
    t2 = time.perf_counter(); pred = model(g, b["x"], b["qf"])
    t3 = time.perf_counter(); loss = info_nce_loss(pred.reshape(bench.N_QUERIES, -1), target)
    t4 = time.perf_counter(); loss.backward()
    t5 = time.perf_counter(); opt.step()
    t6 = time.perf_counter()
    for k, d in zip(seg, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        seg[k] += d
torch.cuda.synchronize()
print("host ms per step:", {k: round(v / n * 1e3, 3) for k, v in seg.items()})
