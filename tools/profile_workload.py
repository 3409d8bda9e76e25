#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes: a known-size device copy (calibrates FETCH_SIZE / WRITE_SIZE, which on gfx950 under-report
wide streaming reads by 2x -- MI355X_MICROARCH.md HBM section) followed by the hot path of one BASELINE config:
    TXE_PROF_WORKLOAD = pgat | pgcn | pgat2   training steps of bench.py --workload <w> (same model / batches)
                      = infer                 MAG-Full all-candidate inference: encode 356 k egonets, fused score + rank and fused score +
                                              best-5 of 2,048 queries"""
import os
import sys

import torch

torch.autograd.set_multithreading_enabled(False)     # as bench.py: the backward runs on the calling thread

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import synthetic as syn  # noqa: E402

W = os.environ.get("TXE_PROF_WORKLOAD", "pgat")
steps = int(os.environ.get("TXE_PROF_STEPS", "4"))
dev = torch.device("cuda:0")
torch.manual_seed(47)


def calibrate():
    # 1 GiB read + 1 GiB write by one elementwise copy kernel (far larger than the 256 MiB Infinity Cache)
    src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    dst.copy_(src)
    torch.cuda.synchronize()
    print("CALIB_BYTES", src.numel() * 4)


if W == "infer":
    from taxoexpan_amd import graph as G
    from taxoexpan_amd.scoring import encode_candidates, rank_all_fused, topk_parents_fused
    tax = syn.make_named_taxonomy("mag_full", seed=47)
    model = bench.make_model("pgat", dev).eval()
    cand, _val, test = syn.split_candidates(tax)
    test = test[:2048]
    dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
    g = G.device_egonet_batch(dtax, cand, seed=7, with_features="lazy")
    queries = tax.features[torch.from_numpy(test)].to(dev)
    pos_off, pos_idx = bench._positives(tax, cand, test)
    with torch.no_grad():
        for _ in range(2):
            hg = encode_candidates(model, g)
        rank_all_fused(model.match, hg, queries, pos_off, pos_idx)
        torch.cuda.synchronize()
        calibrate()
        for _ in range(max(steps // 2, 1)):
            hg = encode_candidates(model, g)
            rank_all_fused(model.match, hg, queries, pos_off, pos_idx)
            topk_parents_fused(model.match, hg, queries, None, 5, True)          # infer.py:96-106: the 5 best parents, no score matrix
    torch.cuda.synchronize()
    print("N", int(g.number_of_nodes()), "E", int(g.number_of_edges()), "G", len(cand), "Q", len(test))
else:
    from taxoexpan_amd.optim import Adam
    tax = syn.make_named_taxonomy("mag_full" if W == "pgat2" else "mag_cs", seed=47)
    model = bench.make_model(W, dev)
    opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
    batches = bench.build_batches(tax, 2, 1000, dev)
    target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
    for i in range(3):
        bench.train_step(model, opt, batches[i % 2], target, 1)
    torch.cuda.synchronize()
    calibrate()
    for i in range(steps):
        bench.train_step(model, opt, batches[i % 2], target, 1)
    torch.cuda.synchronize()
    print("N", [b["n_nodes"] for b in batches], "E", [b["n_edges"] for b in batches])
