#!/usr/bin/env python3
"""Per-kernel average durations of the training step from libtxe's own HIP-event profiler (the facility behind bench.py's
roofline block) -- the quick A/B view between rocprofv3 passes:
    python tools/kernel_times.py [--workload pgat|pgcn|pgat2] [--steps 12] [--filter substr]
Prints the timed step (un-instrumented, ms) and one line per kernel: launches per step, average us, total us per step."""
import argparse
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import synthetic as syn  # noqa: E402
from taxoexpan_amd.optim import Adam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pgat")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(47)
    tax = syn.make_named_taxonomy("mag_full" if a.workload == "pgat2" else "mag_cs", seed=47)
    model = bench.make_model(a.workload, dev)
    opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
    batches = bench.build_batches(tax, 4, 1000, dev)
    target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
    for i in range(10):
        bench.train_step(model, opt, batches[i % 4], target, 1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(50):
        bench.train_step(model, opt, batches[i % 4], target, 1)
    torch.cuda.synchronize()
    print(f"step {(time.perf_counter() - t) / 50 * 1e3:.4f} ms")
    tot, cnt = defaultdict(float), defaultdict(int)
    for i in range(a.steps):
        for name, sec, _work, _kind in bench.profile_step(model, opt, batches[i % 4], target):
            tot[name] += sec
            cnt[name] += 1
    for name in sorted(tot, key=lambda n: -tot[n]):
        if a.filter in name:
            print(f"{cnt[name] / a.steps:5.1f} x {tot[name] / cnt[name] * 1e6:8.1f} us = {tot[name] / a.steps * 1e6:8.1f} us/step  {name}")
    print(f"sum {sum(tot.values()) / a.steps * 1e6:.1f} us/step in {sum(cnt.values()) / a.steps:.1f} profiled launches")


if __name__ == "__main__":
    main()
