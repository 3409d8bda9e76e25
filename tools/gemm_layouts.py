"""NT / NN / TN variants of the MFMA GEMM at the training shapes (txe_gemm_plain), with and without tail splitting."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import _lib
dev = torch.device("cuda:0")
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
wsb = _lib.call("txe_gemm_tail_ws_bytes"); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
def run(layout, M, N, K, splits=1, tail=True):
    if layout == 0: A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); lda, ldb = K, K
    elif layout == 1: A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); lda, ldb = K, N
    else: A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); lda, ldb = M, N
    C = torch.empty(splits * M, N, device=dev)
    f = lambda: _lib.call("txe_gemm_plain", layout, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), N, M, N, K, splits, ws.data_ptr() if tail else None, wsb, _lib.stream_ptr())
    dt = bench(f)
    print(f"{['NT','NN','TN'][layout]} M={M} N={N} K={K} splits={splits} tail={tail}: {dt*1e6:.0f}us {2.0*M*N*K/dt/1e12:.1f}TF")
for lay in (0, 1):
    run(lay, 16384, 2048, 512, tail=False)
    run(lay, 17982, 2080, 512)
    run(lay, 17982, 2048, 512)
for lay in (0, 1, 2):
    run(lay, 4096, 4096, 4096, tail=False)
for s in (1, 4, 8, 16):
    run(2, 512, 2080, 17982, splits=s, tail=False)
    run(2, 2048, 320, 17982, splits=s, tail=False)
