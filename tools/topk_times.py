#!/usr/bin/env python3
"""Times of the scoring loop's three consumers on random graph vectors of the MAG-CS and MAG-Full candidate counts: materialised scores
(txe_score_block), fused score + rank counts, fused score + best-5 (txe_score_topk_block + merge):  gpurun -- python tools/topk_times.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxoexpan_amd import model_zoo as mz, ops  # noqa: E402
from taxoexpan_amd.scoring import rank_all_fused, score_all, topk_parents_fused  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


for name, G, Q in (("mag_cs", 24736, 2459), ("mag_full", 355808, 8192)):
    gen = torch.Generator().manual_seed(1)
    hg = (torch.randn(G, 500, generator=gen) * 0.3).to(dev)
    queries = torch.nn.functional.normalize(torch.randn(Q, 250, generator=gen), dim=1).to(dev)
    torch.manual_seed(5)
    match = mz.LBM(500, 250).to(dev)
    rs = np.random.RandomState(0)
    npos = rs.randint(1, 3, size=Q)
    pos_off = np.concatenate([[0], np.cumsum(npos)])
    pos_idx = np.concatenate([rs.choice(G, size=k, replace=False) for k in npos])
    with torch.no_grad():
        pairs = float(G) * Q
        if G * Q * 4 < 20e9:
            S = score_all(match, hg, queries)
            ts = t(lambda: score_all(match, hg, queries, out=S))
            print(f"{name}: score_all {ts * 1e3:.3f} ms = {pairs / ts / 1e9:.1f} G pairs/s ({500 * pairs / ts / 157.3e12:.3f} of the MFMA roof)")
            del S
        tr = t(lambda: rank_all_fused(match, hg, queries, pos_off, pos_idx))
        print(f"{name}: fused score + rank {tr * 1e3:.3f} ms = {pairs / tr / 1e9:.1f} G pairs/s")
        for k in (1, 5, 8):
            tk = t(lambda: topk_parents_fused(match, hg, queries, None, k, True))
            print(f"{name}: fused score + best-{k} {tk * 1e3:.3f} ms = {pairs / tk / 1e9:.1f} G pairs/s, {Q / tk:.0f} queries/s")
