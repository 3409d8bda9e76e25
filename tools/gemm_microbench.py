#!/usr/bin/env python3
"""Plain-operand GEMM micro-benchmark of libtxe's fp32 MFMA kernel against torch.mm (hipBLASLt), on one MI355X."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, flops, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    return dt * 1e6, flops / dt / 1e12


def plain(layout, A, B, M, N, K, splits=1, tail=False):
    from taxoexpan_amd import _lib
    C = torch.empty((splits, M, N), device=dev)
    wsb = _lib.call("txe_gemm_tail_ws_bytes") if tail else 0
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    def f():
        _lib.call("txe_gemm_plain", layout, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), N, M, N, K, splits,
                  ws.data_ptr() if tail else None, wsb, _lib.stream_ptr())
    f.C = C
    return f


for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 1024), (16384, 2048, 2048), (18000, 2008, 300), (18000, 508, 2050), (18000, 2050, 508)]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(K, N, device=dev)
    Bt = B.t().contiguous()
    us, tf = bench(lambda: ops.bilinear_project(A, B.unsqueeze(0)), 2.0 * M * N * K)
    us2, tf2 = bench(lambda: ops.score_block(A, Bt, False), 2.0 * M * N * K)
    us3, tf3 = bench(lambda: torch.mm(A, B), 2.0 * M * N * K)
    At = A.t().contiguous()
    us4, tf4 = bench(plain(2, At, B, M, N, K), 2.0 * M * N * K)
    ft = plain(0, A, Bt, M, N, K, tail=True)
    us5, tf5 = bench(ft, 2.0 * M * N * K)
    err = (ft.C[0] - torch.mm(A, B)).abs().max().item()
    print(f"   NT with tail splitting: {us5:.0f}us {tf5:.1f}TF  max|err| vs torch.mm {err:.2e}")
    print(f"M={M} N={N} K={K}: NN {us:.0f}us {tf:.1f}TF | NT {us2:.0f}us {tf2:.1f}TF | TN {us4:.0f}us {tf4:.1f}TF | torch.mm {us3:.0f}us {tf3:.1f}TF")
for (M, N, K, S) in [(508, 2050, 18000, 15), (2008, 300, 18000, 21), (512, 2048, 16384, 15)]:
    A = torch.randn(K, M, device=dev)
    B = torch.randn(K, N, device=dev)
    us, tf = bench(plain(2, A, B, M, N, K, S), 2.0 * M * N * K)
    us3, tf3 = bench(lambda: torch.mm(A.t(), B), 2.0 * M * N * K)
    print(f"TN split-K M={M} N={N} K={K} S={S}: {us:.0f}us {tf:.1f}TF | torch.mm(A.t(),B) {us3:.0f}us {tf3:.1f}TF")
