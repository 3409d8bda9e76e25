import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from taxoexpan_amd import ops, _lib
from taxoexpan_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda:0")
def exc(x):
    t = x.abs().double()
    return int(((t != 0) & ((t < 2.0**-100) | (t >= 2.0**120)) | ~torch.isfinite(x)).sum())
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
g = torch.Generator().manual_seed(1)
for (M, N, K) in [(1024, 24736, 500), (17877, 2008, 300)]:
    A = torch.randn(M, K, generator=g).to(dev); B = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    s = stream_ptr()
    Ap = torch.empty(call("txe_split_packed_bytes", M, K), dtype=torch.uint8, device=dev)
    Bp = torch.empty(call("txe_split_packed_bytes", N, K), dtype=torch.uint8, device=dev)
    C = torch.empty(M, N, device=dev)
    call("txe_split_pack", ptr(A), K, M, K, 0, ptr(Ap), s); call("txe_split_pack", ptr(B), K, N, K, 1, ptr(Bp), s)
    t = timed(lambda: call("txe_gemm_nt_split", ptr(Ap), ptr(Bp), M, N, K, ptr(C), N, s))
    print(f"NT {M}x{N}x{K}: {t:.1f} us = {2.0*M*N*K/t*1e-6:.1f} TF/s; exc A {exc(A)} B {exc(B)}; C finite {bool(torch.isfinite(C).all())}")
    marks = (Ap.view(torch.int16).view(-1, 3, 512)[:, 2, 0] == 0x7FC0).sum().item()
    print("  raw fragments in A:", marks)
n, M, N, S = 17877, 2048, 320, 16
A = (torch.randn(n, M, generator=g) * 0.01).to(dev); B = torch.randn(n, N, generator=g).to(dev)
Bt = torch.empty(call("txe_split_packed_t_bytes", n, N), dtype=torch.uint8, device=dev)
ks = (((n + S - 1) // S) + 15) // 16 * 16
part = torch.empty(S, M, N, device=dev)
s = stream_ptr()
call("txe_split_pack_t", ptr(B), N, n, N, ptr(Bt), s)
t = timed(lambda: call("txe_gemm_tn_split", ptr(A), M, M, ptr(Bt), N, n, S, ks, ptr(part), N, M * N, s))
print(f"TN: {t:.1f} us; exc A {exc(A)} B {exc(B)}")
A[:, 100:] = 0
t = timed(lambda: call("txe_gemm_tn_split", ptr(A), M, M, ptr(Bt), N, n, S, ks, ptr(part), N, M * N, s))
print(f"TN with zero columns: {t:.1f} us")
x = torch.tensor([0.0, 1.0, 1e-40, -0.0, 2.0**-100, 2.0**-101], device=dev)
print("frexp exps", torch.frexp(x)[1].tolist())
