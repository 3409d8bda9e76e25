#!/usr/bin/env python3
"""The scoring loop's two routes on one shape -- materialise the [queries x candidates] block then rank it, or score-and-count fused --
wall time (median of 7) and the library profiler's per-kernel view of one pass:  python tools/score_rank_times.py [mag_cs|mag_full] [n_queries]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import _lib, ops, synthetic as syn  # noqa: E402
from taxoexpan_amd.scoring import rank_all_fused, score_all  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "mag_cs"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy(shape, seed=47)
cand, _v, test = syn.split_candidates(tax)
if nq:
    test = test[:nq]
torch.manual_seed(47)
model = bench.make_model("pgat", dev).eval()
hg = torch.randn(len(cand), 500, device=dev) * 0.3
queries = tax.features[torch.from_numpy(test)].to(dev)
off_np, idx_np = bench._positives(tax, cand, test)
off, idx = torch.tensor(off_np, dtype=torch.int32), torch.tensor(idx_np, dtype=torch.int32)
lib = _lib.load()


def med(fn, n=7):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return sorted(ts)[n // 2] * 1e3


def prof(fn):
    lib.txe_profile_reset(); lib.txe_profile_enable(1)
    fn(); torch.cuda.synchronize()
    lib.txe_profile_enable(0)
    buf = ctypes.create_string_buffer(64); ms, work, kind = ctypes.c_float(), ctypes.c_double(), ctypes.c_int()
    tot = 0.0
    for i in range(lib.txe_profile_count()):
        lib.txe_profile_get(i, buf, 64, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(kind))
        tot += ms.value
        print("   %9.1f us  %7.2f %s  %s" % (ms.value * 1e3, work.value / max(ms.value * 1e-3, 1e-9) / 1e12, "TB/s" if kind.value else "TF/s", buf.value.decode()))
    print("   kernels total %.3f ms" % tot)
    lib.txe_profile_reset()


with torch.no_grad():
    S = score_all(model.match, hg, queries)
    print(shape, "candidates", len(cand), "queries", len(test), "positives", len(idx_np))
    print("materialise: score %.3f ms + rank %.3f ms" % (med(lambda: score_all(model.match, hg, queries, out=S)), med(lambda: ops.rank_block(S, off, idx, True))))
    prof(lambda: (score_all(model.match, hg, queries, out=S), ops.rank_block(S, off, idx, True)))
    print("fused: %.3f ms" % med(lambda: rank_all_fused(model.match, hg, queries, off_np, idx_np)))
    prof(lambda: rank_all_fused(model.match, hg, queries, off_np, idx_np))
    r1, r2 = ops.rank_block(S, off, idx, True), rank_all_fused(model.match, hg, queries, off_np, idx_np)
    print("equal ranks", bool(torch.equal(r1.cpu(), r2.cpu())))
