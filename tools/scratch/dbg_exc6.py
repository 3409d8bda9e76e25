import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, numpy as np
from taxoexpan_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
g = torch.Generator().manual_seed(1)
n, M, N, S = 17877, 2048, 320, 16
B = torch.randn(n, N, generator=g).to(dev)
Bt = torch.empty(call("txe_split_packed_t_bytes", n, N), dtype=torch.uint8, device=dev)
ks = (((n + S - 1) // S) + 15) // 16 * 16
part = torch.empty(S, M, N, device=dev)
s = stream_ptr()
call("txe_split_pack_t", ptr(B), N, n, N, ptr(Bt), s)
for scale in (1e-2, 1e-20, 1e-30, 1e-36, 1e-38, 1e-39, 1e-42):
    A = (torch.randn(n, M, generator=g) * scale).to(dev)
    t = timed(lambda: call("txe_gemm_tn_split", ptr(A), M, M, ptr(Bt), N, n, S, ks, ptr(part), N, M * N, s))
    ref = A[:ks].double().t() @ B[:ks].double()
    err = ((part[0].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-300)).item()
    print(f"TN A scale {scale:g}: {t:.1f} us finite={bool(torch.isfinite(part).all())} rel err {err:.2e}  nonzero frac {(part[0] != 0).float().mean().item():.3f}")
# NT
M2, N2, K2 = 2048, 2048, 320
Bm = torch.randn(N2, K2, generator=g).to(dev)
Bp = torch.empty(call("txe_split_packed_bytes", N2, K2), dtype=torch.uint8, device=dev)
call("txe_split_pack", ptr(Bm), K2, N2, K2, 1, ptr(Bp), s)
C = torch.empty(M2, N2, device=dev)
for scale in (1e-2, 1e-30, 1e-38, 1e-39, 1e-42):
    A = (torch.randn(M2, K2, generator=g) * scale).to(dev)
    Ap = torch.empty(call("txe_split_packed_bytes", M2, K2), dtype=torch.uint8, device=dev)
    call("txe_split_pack", ptr(A), K2, M2, K2, 0, ptr(Ap), s)
    t = timed(lambda: call("txe_gemm_nt_split", ptr(Ap), ptr(Bp), M2, N2, K2, ptr(C), N2, s))
    ref = A.double() @ Bm.double().t()
    err = ((C.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-300)).item()
    print(f"NT A scale {scale:g}: {t:.1f} us finite={bool(torch.isfinite(C).all())} rel err {err:.2e}")
