import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from taxoexpan_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
g = torch.Generator().manual_seed(1)
for r in (500, 512, 250):
    hg = torch.randn(24736, r, generator=g).to(dev) * 0.1
    W = torch.randn(1, r, r, generator=g).to(dev) * 0.05
    U = ops.bilinear_project(hg, W)
    Q = torch.randn(1024, r, generator=g).to(dev)
    for ex in (False, True):
        t = timed(lambda: ops.score_block(Q, U, ex))
        print(f"score_block r={r} exp={ex}: {t:.1f} us = {1024*24736/t*1e-3:.1f} G pairs/s; U stride {U.stride()}")
    U2 = torch.randn(24736, r, generator=g).to(dev) * 0.05
    t = timed(lambda: ops.score_block(Q, U2, True))
    print(f"  plain U: {t:.1f} us")
