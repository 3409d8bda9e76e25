import os, sys, time, torch
os.environ["TXE_GEMM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import _lib
dev = torch.device("cuda:0")
M, N, K = 16384, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 320
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
tr = torch.zeros(64 * 48, dtype=torch.int64, device=dev)
f = lambda: _lib.call("txe_gemm_plain", 0, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, 1, tr.data_ptr(), tr.numel() * 8, _lib.stream_ptr())
for _ in range(3): f()
torch.cuda.synchronize()
tr.zero_(); f(); torch.cuda.synchronize()
t = tr.cpu().view(-1, 48)
base = None
for r in t:
    n = int(r[0])
    if n == 0: continue
    ts = r[2:2 + n].tolist()
    if base is None: base = min(int(x[2]) for x in t if int(x[0]) > 0)
    d = [ts[i + 1] - ts[i] for i in range(n - 1)]
    print(f"blk {int(r[1]):5d} start+{(ts[0]-base)/100:8.1f}us  prologue {d[0]/100:6.2f}  ktiles " + " ".join(f"{x/100:5.2f}" for x in d[1:-2]) + f"  last {d[-2]/100:5.2f}  epi {d[-1]/100:5.2f}  total {(ts[-1]-ts[0])/100:6.2f}us")
