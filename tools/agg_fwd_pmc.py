#!/usr/bin/env python3
"""The forward message/reduce sweep of the training batch with one and two destination nodes per wave, ten launches each -- run under
`rocprofv3 --pmc FETCH_SIZE` / `TCC_HIT_sum TCC_MISS_sum` to see what the L2 fetches per variant (kernel names differ by NPW)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import _lib, synthetic as syn  # noqa: E402
from taxoexpan_amd._lib import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
b = bench.build_batches(tax, 1, 1000, dev)[0]
csr = b["g"].csr(dev)
N, E = csr.n_nodes, csr.n_edges
H, D, kp = 4, 500, 2080
torch.manual_seed(0)
ft = torch.randn(N, 2048, device=dev)
out = torch.zeros(N, kp, device=dev)
alpha = torch.empty(E * H, device=dev)
wa = torch.randn(2, kp, device=dev)
mask = torch.randint(0, 2 ** 31, (N, kp // 32), device=dev, dtype=torch.int32)
nxa = torch.empty(N, 2, device=dev)
st = _lib.stream_ptr()
big = torch.empty(1 << 28, device=dev)
for npw in (1, 2):
    for _ in range(10):
        big.fill_(1.0)                                   # (flush the caches between launches)
        call("txe_gat_aggregate_fwd", ptr(csr.rowptr_in), ptr(csr.col_src), N, ptr(ft), 2048, ptr(ft) + 4 * 2000, ptr(ft) + 4 * 2004, 2048, H, D, 0.2,
             0.1, 12345, 1, 0.01, ptr(out), kp, ptr(alpha), ptr(wa), kp, ptr(mask), 0.1, ptr(nxa), npw, st)
torch.cuda.synchronize()
print("N", N, "E", E, "rows MB", N * 8192 / 1e6, "edge rows MB", E * 8192 / 1e6)
