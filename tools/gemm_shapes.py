import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import _lib
dev = torch.device("cuda:0")
def bench(fn, flops, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    return dt * 1e6, flops / dt / 1e12
wsb = _lib.call("txe_gemm_tail_ws_bytes"); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
def run(M, N, K, lda=None, ldb=None, tail=False):
    lda = lda or K; ldb = ldb or K
    A = torch.randn(M, lda, device=dev); B = torch.randn(N, ldb, device=dev); C = torch.empty(M, N, device=dev)
    f = lambda: _lib.call("txe_gemm_plain", 0, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), N, M, N, K, 1, ws.data_ptr() if tail else None, wsb, _lib.stream_ptr())
    us, tf = bench(f, 2.0 * M * N * K)
    print(f"NT M={M} N={N} K={K} lda={lda} ldb={ldb} tail={tail}: {us:.0f}us {tf:.1f}TF")
for args in [(18000,512,2048), (18000,512,2048,2052,2052), (18000,512,2048,2050,2050), (18000,508,2050,2052,2052), (18000,508,2050), (16384,512,2048), (65536,512,2048), (18000,2048,2048)]:
    run(*args); run(*args, tail=True)
