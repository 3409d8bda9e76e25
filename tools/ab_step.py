#!/usr/bin/env python3
"""A/B of the training step inside ONE process (boxes differ by 2-5 %, so two bench.py runs cannot resolve a 1 % change):
    gpurun -- python tools/ab_step.py ops._NO_FUSED_LOGITS [workload]
alternates the named module attribute False / True over groups of steps on bench.py's resident batches and prints both medians."""
import importlib
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from taxoexpan_amd import synthetic as syn  # noqa: E402
from taxoexpan_amd.optim import Adam  # noqa: E402

spec = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "pgat"
mod_name, attr = spec.rsplit(".", 1)
mod = importlib.import_module("taxoexpan_amd." + mod_name)
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
tax = syn.make_named_taxonomy("mag_full" if workload == "pgat2" else "mag_cs", seed=47)
torch.manual_seed(47)
model = bench.make_model(workload, dev)
opt = Adam(model.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
batches = bench.build_batches(tax, 4, seed0=1000, device=dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
it = iter(range(10 ** 9))


def group(n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        bench.train_step(model, opt, batches[next(it) % 4], target, 1)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


res = {False: [], True: []}
for v in (False, True):
    setattr(mod, attr, v)
    group(15)
for rep in range(6):
    for v in (False, True) if rep % 2 == 0 else (True, False):
        setattr(mod, attr, v)
        group(5)
        res[v].append(group())
a, b = float(np.median(res[False])), float(np.median(res[True]))
print(f"{spec} = False: {a:.4f} ms/step   True: {b:.4f} ms/step   (True - False = {1e3 * (b - a):+.1f} us)   groups: "
      f"{[round(x, 4) for x in res[False]]} vs {[round(x, 4) for x in res[True]]}")
