#!/usr/bin/env python3
"""per-kernel HIP-event timings of the bench step (bench.profile_step over the resident batches), top N -- a quick look between builds
usage: python tools/kt_quick.py [workload] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from taxoexpan_amd import synthetic as syn
from taxoexpan_amd.optim import Adam
wl = sys.argv[1] if len(sys.argv) > 1 else "pgat"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
if os.environ.get("TXE_NO_WALK_PLAN", "0") == "1":
    from taxoexpan_amd import ops as _o
    _o._NO_WALK_PLAN = True
if os.environ.get("TXE_NO_VIRTUAL_X", "0") == "1":
    from taxoexpan_amd import ops as _o2
    _o2._NO_VIRTUAL_X = True
if os.environ.get("TXE_FWD_SWEEP"):
    from taxoexpan_amd import ops
    ops._FWD_SWEEP = int(os.environ["TXE_FWD_SWEEP"])
torch.autograd.set_multithreading_enabled(False)
if wl == "semeval":
    bench.N_QUERIES = 64
tax = syn.make_named_taxonomy({"pgat2": "mag_full", "semeval": "semeval_noun"}.get(wl, "mag_cs"), seed=47)
torch.manual_seed(47)
model = bench.make_model(wl, dev)
opt = Adam(model.parameters(), lr=bench.LR, weight_decay=0, amsgrad=True)
batches = bench.build_batches(tax, 4, seed0=1000, device=dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
for i in range(300):
    bench.train_step(model, opt, batches[i % 4], target, 1)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(200):
    loss = bench.train_step(model, opt, batches[i % 4], target, 1)
torch.cuda.synchronize()
print(f"{wl}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step, loss {float(loss):.3f}")
recs = [bench.profile_step(model, opt, batches[i % 4], target) for i in range(12)]
agg = {}
for rec in recs:
    for name, dt, work, kind in rec:
        a = agg.setdefault(name, [0.0, 0, work, kind]); a[0] += dt; a[1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values()) / 12
print(f"sum of kernel times {tot * 1e6:.1f} us/step over {sum(v[1] for v in agg.values()) // 12} launches")
for name, (t, n, work, kind) in rows[:top]:
    us = t / n * 1e6
    rate = work / (t / n)
    print(f"  {name[:52]:52s} {us:8.1f} us x{n // 12}  {'%.2f TB/s' % (rate / 1e12) if kind == 1 else '%.1f TF/s' % (rate / 1e12)}")
