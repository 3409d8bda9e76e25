#!/usr/bin/env python3
"""txe_gemm_nt_split (fp32 products on the bf16 matrix pipe) against float64 and against the fp32-MFMA route, with timings.
usage: python tools/split_gemm_probe.py [M N K]      (default: the training step's first-layer projection 17877 x 2008 x 300)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd._lib import call, ptr, stream_ptr  # noqa: E402


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (17877, 2008, 300)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(M, K, generator=g).to(dev)
    B = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    # an asymmetric corner: exact small integers and a few huge / tiny magnitudes
    A[:7, :5] = torch.arange(35, dtype=torch.float32).reshape(7, 5).to(dev)
    A[7, :] *= 1e18
    A[8, :] *= 1e-18
    s = stream_ptr()
    na, nb = call("txe_split_packed_bytes", M, K), call("txe_split_packed_bytes", N, K)
    Ap = torch.empty(na, dtype=torch.uint8, device=dev)
    Bp = torch.empty(nb, dtype=torch.uint8, device=dev)
    C = torch.full((M, N), float("nan"), device=dev)
    call("txe_split_pack", ptr(A), K, M, K, 0, ptr(Ap), s)
    call("txe_split_pack", ptr(B), K, N, K, 1, ptr(Bp), s)
    call("txe_gemm_nt_split", ptr(Ap), ptr(Bp), M, N, K, ptr(C), N, s)
    torch.cuda.synchronize()
    ref = A.double() @ B.double().t()
    c32 = A @ B.t()
    scale = (A.double().abs() @ B.double().abs().t()).clamp_min(1e-300)       # |a|.|b|: the natural error scale of a dot product
    e_split = ((C.double() - ref).abs() / scale).max().item()
    e_f32 = ((c32.double() - ref).abs() / scale).max().item()
    rel = lambda x: ((x.double() - ref).norm() / ref.norm()).item()
    print(f"shape {M} x {N} x {K}: max |err| / (|a|.|b|): split {e_split:.3e}  torch fp32 {e_f32:.3e}   rel. Frobenius: split {rel(C):.3e}  fp32 {rel(c32):.3e}")
    assert torch.isfinite(C).all()
    t_pa = timed(lambda: call("txe_split_pack", ptr(A), K, M, K, 0, ptr(Ap), s))
    t_pb = timed(lambda: call("txe_split_pack", ptr(B), K, N, K, 1, ptr(Bp), s))
    t_g = timed(lambda: call("txe_gemm_nt_split", ptr(Ap), ptr(Bp), M, N, K, ptr(C), N, s))
    t_mm = timed(lambda: torch.mm(A, B.t(), out=c32))
    fl = 2.0 * M * N * K
    print(f"pack A {t_pa:.1f} us, pack B {t_pb:.1f} us, product {t_g:.1f} us = {fl / t_g * 1e-6:.1f} TF/s algorithmic "
          f"({6 * fl / t_g * 1e-6 / 2500:.2f} of the bf16 pipe's 2.5 PF/s on six plane products); torch.mm fp32 {t_mm:.1f} us")


def main_tn():
    """weight-gradient form: part[z] = dY[slice z]^T X[slice z]  (the step's first-layer shapes: 17877 nodes, 2048 x 320)"""
    n, M, N, S = 17877, 2048, 320, 16
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(2)
    A = (torch.randn(n, M, generator=g) * 0.01).to(dev)
    B = torch.randn(n, N, generator=g).to(dev)
    B[:, 300:] = 0
    s = stream_ptr()
    Bt = torch.empty(call("txe_split_packed_t_bytes", n, N), dtype=torch.uint8, device=dev)
    ksplit = ((n + S - 1) // S + 31) // 32 * 32
    part = torch.full((S, M, N), float("nan"), device=dev)
    call("txe_split_pack_t", ptr(B), N, n, N, ptr(Bt), s)
    call("txe_gemm_tn_split", ptr(A), M, M, ptr(Bt), N, n, S, ksplit, ptr(part), N, M * N, s)
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()
    C = part.double().sum(0)
    ref = A.double().t() @ B.double()
    c32 = (A.t() @ B).double()
    scale = (A.double().abs().t() @ B.double().abs()).clamp_min(1e-300)
    print(f"TN {M} x {N} over {n} rows, {S} slices of {ksplit}: max |err| / (|a|.|b|): split {((C - ref).abs() / scale).max().item():.3e}  "
          f"torch fp32 {((c32 - ref).abs() / scale).max().item():.3e}")
    for z in (0, S - 1):
        lo, hi = z * ksplit, min(n, (z + 1) * ksplit)
        rz = A[lo:hi].double().t() @ B[lo:hi].double()
        sz = (A[lo:hi].double().abs().t() @ B[lo:hi].double().abs()).clamp_min(1e-300)
        print(f"  slice {z}: max rel err {((part[z].double() - rz).abs() / sz).max().item():.3e}")
    t_p = timed(lambda: call("txe_split_pack_t", ptr(B), N, n, N, ptr(Bt), s))
    t_g = timed(lambda: call("txe_gemm_tn_split", ptr(A), M, M, ptr(Bt), N, n, S, ksplit, ptr(part), N, M * N, s))
    out = torch.empty(M, N, device=dev)
    t_mm = timed(lambda: torch.mm(A.t(), B, out=out))
    print(f"pack {t_p:.1f} us, product {t_g:.1f} us = {2.0 * M * N * n / t_g * 1e-6:.1f} TF/s algorithmic (padded shapes); torch.mm fp32 {t_mm:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tn":
        main_tn()
    else:
        main()
