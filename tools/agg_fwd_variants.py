#!/usr/bin/env python3
"""The forward message/reduce sweep of the first PGAT layer on the BASELINE training batch, stand-alone, with its optional parts
switched off one at a time (alpha kept for backward, attention dropout, the next layer's logits epilogue, that layer's dropout mask):
what does each cost?    python tools/agg_fwd_variants.py"""
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
H, D, kp = 4, 500, 2048
torch.manual_seed(0)
ft = torch.randn(N, H * D, device=dev)
a12 = torch.randn(N, 2 * H, device=dev)
out = torch.zeros(N, kp, device=dev)
alpha = torch.empty(E * H, device=dev)
wa = torch.randn(2, kp, device=dev)
mask = torch.randint(0, 2 ** 31, (N, kp // 32), device=dev, dtype=torch.int32)
nxa = torch.empty(N, 2, device=dev)
st = _lib.stream_ptr()


def run(keep_alpha, attn_p, nx, nx_p):
    call("txe_gat_aggregate_fwd", ptr(csr.rowptr_in), ptr(csr.col_src), N, ptr(ft), H * D, ptr(a12), ptr(a12[:, H:]), 2 * H, H, D, 0.2,
         attn_p, 12345, 1, 0.01, ptr(out), kp, ptr(alpha) if keep_alpha else None, ptr(wa) if nx else None, kp,
         ptr(mask) if nx_p > 0 else None, nx_p, ptr(nxa) if nx else None, 0, st)


def t(*a):
    for _ in range(5):
        run(*a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run(*a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


print(f"N {N} E {E}")
for name, a in (("training variant (alpha, attn dropout, logits epilogue, mask)", (True, 0.1, True, 0.5)),
                ("  without alpha kept", (False, 0.1, True, 0.5)),
                ("  without attention dropout", (True, 0.0, True, 0.5)),
                ("  without the next layer's mask", (True, 0.1, True, 0.0)),
                ("  without the logits epilogue", (True, 0.1, False, 0.0)),
                ("eval variant with epilogue", (False, 0.0, True, 0.0)),
                ("plain aggregation", (False, 0.0, False, 0.0))):
    print(f"{name:70s} {t(*a):7.1f} us")
