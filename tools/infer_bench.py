#!/usr/bin/env python3
"""All-candidate inference (test_fast.py:99-225) on a synthetic taxonomy of a named shape, one MI355X:
encode every candidate egonet (one batch, and in `-b` chunks like the reference), then score every test query against
every candidate in query blocks with the ranks taken on device.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import TaxoExpan, ops, synthetic as syn  # noqa: E402
from taxoexpan_amd.scoring import encode_candidates  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="mag_full", choices=list(syn.SHAPES))
ap.add_argument("--chunk", type=int, default=-1, help="-b of test_fast.py: egonets per encoder batch (-1 = one batch)")
ap.add_argument("--qblock", type=int, default=1024)
ap.add_argument("--max-queries", type=int, default=0)
ap.add_argument("--profile", action="store_true", help="per-kernel HIP-event timing of one encode pass (stderr)")
args = ap.parse_args()

dev = torch.device("cuda:0")
if os.environ.get("TXE_FWD_SWEEP"):                     # txe_gat_aggregate_*_fwd's npw argument (4 = one wave per node, 3 = the egonet walk)
    ops._FWD_SWEEP = int(os.environ["TXE_FWD_SWEEP"])
t0 = time.perf_counter()
tax = syn.make_named_taxonomy(args.shape, seed=47)
cand, val, test = syn.split_candidates(tax)
if args.max_queries:
    test = test[:args.max_queries]
t_tax = time.perf_counter() - t0
torch.manual_seed(47)
model = TaxoExpan("PGAT", "WMR", "LBM", **bench.MAG).to(dev).eval()

t0 = time.perf_counter()
chunks = [cand] if args.chunk <= 0 else [cand[i:i + args.chunk] for i in range(0, len(cand), args.chunk)]
graphs = []
for c in chunks:
    g = syn.egonet_batch(tax, c, seed=7)
    g.ndata["x"] = g.ndata["x"].to(dev)
    g.csr(dev)
    graphs.append(g)
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
n_nodes = sum(g.number_of_nodes() for g in graphs)
n_edges = sum(g.number_of_edges() for g in graphs)

# the same batches built on device (txe_egonet_*): taxonomy CSR + feature table resident, anchors in, batched graph out
from taxoexpan_amd import graph as G  # noqa: E402
dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
dgraphs = [G.device_egonet_batch(dtax, c, seed=7) for c in chunks]
torch.cuda.synchronize()
t0 = time.perf_counter()
dgraphs = [G.device_egonet_batch(dtax, c, seed=7) for c in chunks]
torch.cuda.synchronize()
t_dbuild = time.perf_counter() - t0
del dgraphs

hg = encode_candidates(model, graphs)          # warm-up (allocator, code objects)
torch.cuda.synchronize()
t0 = time.perf_counter()
hg = encode_candidates(model, graphs)
torch.cuda.synchronize()
t_enc = time.perf_counter() - t0

# the same candidates as device-built batches whose features stay rows of the taxonomy table: the eval-mode layer-0 projection runs
# once per taxonomy node and is gathered (SURVEY 8f-2 "dedup by _id")
lgraphs = [G.device_egonet_batch(dtax, c, seed=7, with_features="lazy") for c in chunks]
hg_l = encode_candidates(model, lgraphs)
torch.cuda.synchronize()
t0 = time.perf_counter()
hg_l = encode_candidates(model, lgraphs)
torch.cuda.synchronize()
t_enc_dedup = time.perf_counter() - t0
mgraphs = [G.device_egonet_batch(dtax, c, seed=7) for c in chunks]
hg_m = encode_candidates(model, mgraphs)
dedup_err = float((hg_l - hg_m).abs().max() / hg_m.abs().max())
n_edges_dev = sum(g.number_of_edges() for g in lgraphs)
del mgraphs, hg_m

if args.profile:
    import ctypes
    from taxoexpan_amd import _lib
    lib = _lib.load()
    lib.txe_profile_reset()
    lib.txe_profile_enable(1)
    encode_candidates(model, lgraphs if os.environ.get("TXE_PROFILE_DEDUP", "0") == "1" else graphs)
    torch.cuda.synchronize()
    lib.txe_profile_enable(0)
    buf = ctypes.create_string_buffer(64)
    ms, work, kind = ctypes.c_float(), ctypes.c_double(), ctypes.c_int()
    tot = 0.0
    for i in range(lib.txe_profile_count()):
        lib.txe_profile_get(i, buf, 64, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(kind))
        tot += ms.value
        rate = work.value / (ms.value * 1e-3)
        print(f"  {buf.value.decode():44s} {ms.value * 1e3:10.1f} us  " + (f"{rate / 1e12:7.1f} TF/s" if kind.value == 0 else f"{rate / 1e9:7.0f} GB/s"),
              file=sys.stderr)
    print(f"  kernel total {tot:.3f} ms", file=sys.stderr)
    lib.txe_profile_reset()

queries = tax.features[torch.from_numpy(test)].to(dev)
cand_index = np.full(tax.n_nodes, -1, dtype=np.int64)
cand_index[cand] = np.arange(len(cand))
pos_lists = [cand_index[tax.par_idx[tax.par_ptr[q]:tax.par_ptr[q + 1]]] for q in test]
pos_lists = [p[p >= 0] for p in pos_lists]
U = ops.bilinear_project(hg, model.match.W.weight)
S = torch.empty((args.qblock, len(cand)), dtype=torch.float32, device=dev)


def run_scoring():
    all_ranks = []
    for q0 in range(0, len(test), args.qblock):
        qb = queries[q0:q0 + args.qblock]
        Sb = ops.score_block(qb, U, True, out=S[:qb.shape[0]])
        pl = pos_lists[q0:q0 + qb.shape[0]]
        off = torch.tensor(np.concatenate([[0], np.cumsum([len(p) for p in pl])]), dtype=torch.int32)
        idx = torch.tensor(np.concatenate(pl) if pl else np.zeros(0), dtype=torch.int32)
        all_ranks.append(ops.rank_block(Sb, off, idx, True))
    return torch.cat(all_ranks)


run_scoring()
torch.cuda.synchronize()
t0 = time.perf_counter()
ranks = run_scoring()
torch.cuda.synchronize()
t_score = time.perf_counter() - t0
# fused scoring + ranking: no [queries x candidates] block is ever stored
from taxoexpan_amd.scoring import rank_all_fused  # noqa: E402
pos_off_all = np.concatenate([[0], np.cumsum([len(p) for p in pos_lists])])
pos_idx_all = np.concatenate(pos_lists) if pos_lists else np.zeros(0, dtype=np.int64)
ranks_f = rank_all_fused(model.match, hg, queries, pos_off_all, pos_idx_all, block=args.qblock)
torch.cuda.synchronize()
t0 = time.perf_counter()
ranks_f = rank_all_fused(model.match, hg, queries, pos_off_all, pos_idx_all, block=args.qblock)
torch.cuda.synchronize()
t_fused = time.perf_counter() - t0
assert torch.equal(ranks_f.cpu(), ranks.cpu()), "fused ranks differ from the materialised path"
pairs = float(len(cand)) * len(test)
print(json.dumps(dict(shape=args.shape, candidates=int(len(cand)), queries=int(len(test)), nodes=n_nodes, edges=n_edges,
                      encoder_batches=len(graphs), host_taxonomy_s=t_tax, host_egonet_build_and_upload_s=t_build, device_egonet_build_s=t_dbuild,
                      encode_s=t_enc, encode_edges_per_s=n_edges / t_enc, encode_dedup_s=t_enc_dedup,
                      encode_dedup_edges_per_s=n_edges_dev / t_enc_dedup, dedup_max_rel_err=dedup_err, score_and_rank_s=t_score,
                      candidates_scored_per_s=pairs / t_score, fused_score_and_rank_s=t_fused, candidates_scored_per_s_fused=pairs / t_fused, candidates_scored_per_s_incl_encode=pairs / (t_score + t_enc),
                      mean_rank=float(ranks.float().mean()), hbm_gb=torch.cuda.max_memory_allocated() / 1e9)))
