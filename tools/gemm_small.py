"""micro-benchmark of the three 4096 x 500 x 250 bilinear products (DESIGN 9 gap 3): leading dimensions 250 vs 256, tail splitting"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import _lib
dev = torch.device("cuda:0")
def bench(fn, flops, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    return dt * 1e6, flops / dt / 1e12
wsb = _lib.call("txe_gemm_tail_ws_bytes"); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
def run(layout, M, N, K, lda, ldb, ldc, tail=False, splits=1):
    # layout 0: A[M][K] B[N][K]; 1: A[M][K] B[K][N]; 2: A[K][M] B[K][N]
    ra = M if layout < 2 else K; rb = N if layout == 0 else K
    A = torch.randn(ra, lda, device=dev); B = torch.randn(rb, ldb, device=dev); C = torch.empty(splits * M, ldc, device=dev)
    f = lambda: _lib.call("txe_gemm_plain", layout, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc, M, N, K, splits,
                          ws.data_ptr() if tail else None, wsb, _lib.stream_ptr())
    us, tf = bench(f, 2.0 * M * N * K)
    print(f"layout={layout} M={M} N={N} K={K} lda={lda} ldb={ldb} ldc={ldc} tail={tail} splits={splits}: {us:.1f}us {tf:.1f}TF", flush=True)
G = int(os.environ.get("G", 4096))
print("# U = hg W  (nn)")
for ldb, N in [(250, 250), (256, 250), (256, 256)]:
    for tail in (False, True):
        run(1, G, N, 500, 500, ldb, ldb, tail)
print("# d_e1 = dU W^T  (nt)")
for ld, K in [(250, 250), (256, 250), (256, 256)]:
    for tail in (False, True):
        run(0, G, 500, K, ld, ld, 500, tail)
print("# dW = hg^T dU  (tn, split-K)")
for ldb, N in [(250, 250), (256, 250), (256, 256)]:
    for s in (1, 2, 4, 8, 16):
        run(2, 500, N, G, 500, ldb, ldb, False, s)
