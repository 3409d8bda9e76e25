#!/usr/bin/env python3
"""Is the host anywhere on the step's critical path?  30 us of busy-waiting is put in front of ONE C-ABI call per run (the library's
`call`, wrapped) and the BASELINE step re-timed: where the device waits for the host the step grows by ~30 us, where the host runs ahead
it does not move.  Round 5: +1 .. +5 us at every call site -- the 4,096-egonet PGAT step is device-bound from end to end (the host enqueues
it in 0.79 ms under 0.95 ms of kernels); the PGCN step is the opposite (enqueue time = wall time).    gpurun -- python tools/host_slack.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from taxoexpan_amd import synthetic as syn, ops, _lib
from taxoexpan_amd.optim import Adam
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = bench.make_model("pgat", dev)
opt = Adam(model.parameters(), lr=bench.LR, weight_decay=0, amsgrad=True)
batches = bench.build_batches(tax, 4, seed0=1000, device=dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
DELAY = {"name": None, "us": 0.0}
orig_call = _lib.call
def slow_call(name, *args):
    if name == DELAY["name"]:
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < DELAY["us"]:
            pass
    return orig_call(name, *args)
_lib.call = slow_call
ops.call = slow_call
import taxoexpan_amd.loss as L, taxoexpan_amd.optim as O
for m in (L, O):
    if hasattr(m, "call"): m.call = slow_call
def measure():
    for i in range(20):
        bench.train_step(model, opt, batches[i % 4], target, 1)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40):
            bench.train_step(model, opt, batches[i % 4], target, 1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40)
    ts.sort(); return 1e3 * ts[2]
for i in range(300):
    bench.train_step(model, opt, batches[i % 4], target, 1)
import gc; gc.collect(); gc.freeze()
base = measure()
print(f"base {base:.4f} ms")
for name in ("txe_gat_layers_prepare", "txe_gat_dense_fwd_split", "txe_gat_aggregate_fwd", "txe_rows_find_runs", "txe_bilinear_folded_fwd", "txe_gat_collapse_fwd", "txe_gat_collapse_fold_scores",
             "txe_info_nce", "txe_bilinear_folded_bwd", "txe_gat_collapse_bwd_fused", "txe_gat_dense_bwd", "txe_adam_step"):
    DELAY.update(name=name, us=30.0)
    t = measure()
    print(f"+30 us of host time before {name:34s}: {t:.4f} ms  ({1e3 * (t - base):+.1f} us)")
DELAY.update(name=None)
print(f"base again {measure():.4f} ms")
# host enqueue time of one step (no device wait): the queue is drained first, then 20 steps are enqueued back to back
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    bench.train_step(model, opt, batches[i % 4], target, 1)
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
print(f"host enqueue time per step (20 steps enqueued without waiting): {1e3 * t_host:.4f} ms")
