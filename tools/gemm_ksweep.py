"""K / M sweeps of the NT GEMM (txe_gemm_plain) at the forward-projection shapes: time per 128x128 tile vs k-tiles."""
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
def run(M, N, K, tail=False):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    f = lambda: _lib.call("txe_gemm_plain", 0, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, 1, ws.data_ptr() if tail else None, wsb, _lib.stream_ptr())
    dt = bench(f)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"NT M={M} N={N} K={K} tail={tail}: {dt*1e6:.0f}us {2.0*M*N*K/dt/1e12:.1f}TF  tiles={tiles} rounds={tiles/512:.2f} us/round={dt*1e6/max(1,-(-tiles//512)):.1f}")
for K in (32, 64, 128, 320, 640, 1280, 2560):
    run(16384, 2048, K)          # 2048 tiles = exactly 4 rounds of 512 slots
for K in (320, 2048):
    run(512 * 128 // 16, 2048, K)   # 1 round
    run(18048, 2048, K); run(18048, 2048, K, True)
run(16384, 512, 2080); run(18048, 512, 2080); run(18048, 512, 2080, True)
