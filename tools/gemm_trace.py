"""run after tools/gemm_trace_patch.py:  python tools/gemm_trace.py <layout 0|1|2> M N K [splits]"""
import os, sys, torch
os.environ["TXE_GEMM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import _lib
dev = torch.device("cuda:0")
lay, M, N, K = (int(x) for x in sys.argv[1:5])
splits = int(sys.argv[5]) if len(sys.argv) > 5 else 1
if lay == 0: A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); lda, ldb = K, K
elif lay == 1: A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); lda, ldb = K, N
else: A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); lda, ldb = M, N
C = torch.empty(splits * M, N, device=dev)
tr = torch.zeros(64 * 48, dtype=torch.int64, device=dev)
f = lambda: _lib.call("txe_gemm_plain", lay, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), N, M, N, K, splits, tr.data_ptr(), tr.numel() * 8, _lib.stream_ptr())
for _ in range(3): f()
torch.cuda.synchronize(); tr.zero_(); f(); torch.cuda.synchronize()
for r in tr.cpu().view(-1, 48):
    n = int(r[0])
    if n == 0: continue
    ts = r[2:2 + n].tolist()
    d = [ts[i + 1] - ts[i] for i in range(n - 1)]
    print(f"blk {int(r[1]):5d} prologue {d[0]:6d}  k-tiles " + " ".join(f"{x:5d}" for x in d[1:-2][:14]) + f" ... last {d[-2]:5d}  epi {d[-1]:6d}  total {ts[-1]-ts[0]:8d} cycles")
