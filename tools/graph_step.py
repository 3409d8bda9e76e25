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
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(47)
    tax = syn.make_named_taxonomy("mag_full" if a.workload == "pgat2" else "mag_cs", seed=47)
    model = bench.make_model(a.workload, dev)
    opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
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
    print(f"eager  {timeit(lambda: bench.train_step(model, opt, batches[next(it) % 2], target, 1)):.4f} ms/step")
    graphs = []
    for b in batches:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            bench.train_step(model, opt, b, target, 1)          # (warm-up on the capture stream)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            bench.train_step(model, opt, b, target, 1)
        graphs.append(g)
    torch.cuda.synchronize()
    print(f"graph  {timeit(lambda: graphs[next(it) % 2].replay()):.4f} ms/step")


if __name__ == "__main__":
    main()
