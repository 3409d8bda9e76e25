import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import ops
dev = torch.device("cuda:0")
M = N = K = 4096
A = torch.randn(M, K, device=dev); Bt = torch.randn(N, K, device=dev)
for _ in range(3): ops.score_block(A, Bt, False)
torch.cuda.synchronize()
