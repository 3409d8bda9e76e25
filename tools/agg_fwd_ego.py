#!/usr/bin/env python3
"""The forward message/reduce sweep of the training batch: one wave per node (npw 1 | 2) against the egonet walk with several window
sizes, HIP-event times with the caches flushed between launches (the sweep's inputs were written by the projection GEMM long before in a
step, so cold is the honest case) -- and warm."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import _lib, synthetic as syn  # noqa: E402
from taxoexpan_amd._lib import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "mag_cs"
tax = syn.make_named_taxonomy(which, seed=47)
if len(sys.argv) > 2:
    bench.N_QUERIES = int(sys.argv[2])               # queries per batch (x 32 egonets)
b = bench.build_batches(tax, 1, 1000, dev)[0]
csr = b["g"].csr(dev)
N, E = csr.n_nodes, csr.n_edges
H, D, kp = (4, 500, 2080) if which != "semeval" else (4, 600, 2464)
F = H * D
ld = (F + 2 * H + 31) // 32 * 32
torch.manual_seed(0)
ft = torch.randn(N, ld, device=dev)
out = torch.zeros(N, kp, device=dev)
alpha = torch.empty(E * H, device=dev)
wa = torch.randn(2, kp, device=dev)
mask = torch.randint(0, 2 ** 31, (N, kp // 32), device=dev, dtype=torch.int32)
nxa = torch.empty(N, 2, device=dev)
st = _lib.stream_ptr()
big = torch.empty(1 << 28, device=dev)
print("N", N, "E", E, "rows MB", N * 4 * F / 1e6, "alg MB", (N * 4 * F + N * 4 * kp) / 1e6)
ref = None
for npw in (2, 1, 3, 12, 16, 20, 24, 28, 32):
    for cold in (True, False):
        ts = []
        for _ in range(12):
            if cold:
                big.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call("txe_gat_aggregate_fwd", ptr(csr.rowptr_in), ptr(csr.col_src), N, ptr(ft), ld, ptr(ft) + 4 * F, ptr(ft) + 4 * (F + H), ld, H, D, 0.2,
                 0.1, 12345, 1, 0.01, ptr(out), kp, ptr(alpha), ptr(wa), kp, ptr(mask), 0.1, ptr(nxa), npw, st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        print(f"npw {npw:2d} {'cold' if cold else 'warm'}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}")
    cur = (out.clone(), alpha.clone(), nxa.clone())
    if ref is None:
        ref = cur
    else:
        print("   out equal", torch.equal(ref[0], cur[0]), "alpha equal", torch.equal(ref[1], cur[1]), "nx max diff",
              float((ref[2] - cur[2]).abs().max()), "of", float(ref[2].abs().max()))
