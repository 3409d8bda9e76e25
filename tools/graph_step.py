#!/usr/bin/env python3
"""Experiment: the training step of a resident batch captured in a HIP graph (torch.cuda.CUDAGraph) against the eager step.
    python tools/graph_step.py [--workload pgat|pgcn|pgat2]
The capture freezes everything passed by value (dropout seeds, Adam's step count): a measurement of what graph replay would save,
not a training loop."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import synthetic as syn  # noqa: E402
from taxoexpan_amd.optim import Adam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pgat")
    ap.add_argument("--engine-threads", action="store_true", help="leave autograd's per-device engine thread on (the capture then dies)")
    a = ap.parse_args()
    torch.autograd.set_multithreading_enabled(a.engine_threads)       # backward on the calling thread: the thread that captures
    dev = torch.device("cuda:0")
    # EVERYTHING runs on the stream that will capture: parameters' AccumulateGrad nodes outlive an iteration (g.ndata['h'] of a resident
    # batch keeps the last step's autograd graph alive) and keep the stream they were made on -- made on the default stream they would
    # drag the legacy stream into the capture
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        run(a, dev, s)


def run(a, dev, s):
    torch.manual_seed(47)
    tax = syn.make_named_taxonomy("mag_full" if a.workload == "pgat2" else "mag_cs", seed=47)
    model = bench.make_model(a.workload, dev)
    opt = Adam(model.parameters(), lr=bench.LR, weight_decay=0, amsgrad=True)
    batches = bench.build_batches(tax, 2, 1000, dev)
    target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
    for i in range(10):
        bench.train_step(model, opt, batches[i % 2], target, 1)
    torch.cuda.synchronize()

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    it = iter(range(10 ** 9))
    print(f"eager  {timeit(lambda: bench.train_step(model, opt, batches[next(it) % 2], target, 1)):.4f} ms/step", flush=True)
    graphs = []
    for b in batches:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            bench.train_step(model, opt, b, target, 1)
        graphs.append(g)
        print("captured", flush=True)
    torch.cuda.synchronize()
    print(f"graph  {timeit(lambda: graphs[next(it) % 2].replay()):.4f} ms/step", flush=True)


if __name__ == "__main__":
    main()
