#!/usr/bin/env python3
"""One micro-benchmark for libtxe's fp32 MFMA GEMM (txe_gemm_plain) on one MI355X -- replaces the round-1 one-off scripts.

    python tools/gemm_bench.py model                       # every dense product of the MAG training step, by name
    python tools/gemm_bench.py ksweep                      # NT 16384 x 2048 x K, K = 32 .. 2560 (time per round of 512 tiles)
    python tools/gemm_bench.py LAYOUT M N K [--lda ..] [--ldb ..] [--ldc ..] [--splits S] [--tail] [--mm]
LAYOUT 0: C = A[M][K] B[N][K]^T   1: C = A[M][K] B[K][N]   2: C = A[K][M]^T B[K][N].   --mm also times torch.mm (hipBLASLt)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


_ws = None


def run(layout, M, N, K, lda=None, ldb=None, ldc=None, splits=1, tail=False, mm=False, label="", route=0):
    global _ws
    if _ws is None:
        _ws = torch.empty(_lib.call("txe_gemm_tail_ws_bytes"), dtype=torch.uint8, device=dev)
    ra, ca = (M, K) if layout < 2 else (K, M)
    rb, cb = (N, K) if layout == 0 else (K, N)
    lda, ldb, ldc = lda or ca, ldb or cb, ldc or N
    A = torch.randn(ra, lda, device=dev)
    B = torch.randn(rb, ldb, device=dev)
    C = torch.empty(splits * M, ldc, device=dev)
    f = lambda: _lib.call("txe_gemm_plain", layout, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc, M, N, K, splits, route,
                          _ws.data_ptr() if tail else None, _ws.numel() if tail else 0, _lib.stream_ptr())
    dt = timeit(f)
    flops = 2.0 * M * N * K
    a, b = A[:, :ca], B[:, :cb]
    ref = (a if layout < 2 else a.t()).double() @ (b.t() if layout == 0 else b).double()
    got = C.view(splits, M, ldc)[:, :, :N].double().sum(0)
    err = float((got - ref).abs().max() / ref.abs().max())
    line = f"{label:28s} {['NT', 'NN', 'TN'][layout]} M={M} N={N} K={K} splits={splits} tail={int(tail)}: {dt * 1e6:7.1f} us {flops / dt / 1e12:6.1f} TF/s  rel.err {err:.1e}"
    if mm:
        dt2 = timeit(lambda: torch.mm(a if layout < 2 else a.t(), b.t() if layout == 0 else b))
        line += f" | torch.mm {dt2 * 1e6:7.1f} us {flops / dt2 / 1e12:6.1f} TF/s"
    print(line, flush=True)
    return dt


def model_shapes(n=17877, g=4096):
    """the dense products of one MAG PGAT+WMR+LBM training step (padded operands as the library lays them out)"""
    return [("L0 fwd  Y = X Wp^T", 0, n, 2008, 320, {}), ("folded hg = Z Wp^T", 0, g, 500, 2080, {}),
            ("folded dZ = d_hg Wp", 1, g, 2080, 500, dict(ldb=2080)), ("folded dW = d_hg^T Z", 2, 500, 2080, g, dict(splits=4)),
            ("L0 dX (pos cols)", 1, n, 72, 2048, dict(ldb=320, ldc=320)), ("L0 dW = d_Y^T X", 2, 2048, 320, n, dict(splits=8)),
            ("LBM U = hg W", 1, g, 250, 500, {}), ("LBM d_e1 = dU W^T", 0, g, 500, 250, {}), ("LBM dW = hg^T dU", 2, 500, 250, g, dict(splits=8))]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "model":
        for label, lay, M, N, K, kw in model_shapes():
            run(lay, M, N, K, tail=kw.get("splits", 1) == 1, label=label, **kw)
    elif len(sys.argv) > 1 and sys.argv[1] == "ksweep":
        for K in (32, 64, 128, 320, 640, 1280, 2560):
            run(0, 16384, 2048, K)
    else:
        ap = argparse.ArgumentParser()
        ap.add_argument("layout", type=int)
        ap.add_argument("M", type=int)
        ap.add_argument("N", type=int)
        ap.add_argument("K", type=int)
        for o in ("--lda", "--ldb", "--ldc"):
            ap.add_argument(o, type=int, default=None)
        ap.add_argument("--splits", type=int, default=1)
        ap.add_argument("--tail", action="store_true")
        ap.add_argument("--mm", action="store_true")
        ap.add_argument("--route", type=int, default=0)
        a = ap.parse_args()
        run(a.layout, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.splits, a.tail, a.mm, route=a.route)
