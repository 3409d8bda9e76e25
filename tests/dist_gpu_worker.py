"""TEST INFRASTRUCTURE: one rank of a world-size-N job whose ranks all sit on cuda:0 (a single-GPU box) over gloo, launched by
tests/test_gpu_eval_infer.py through torch.distributed.run.  Every rank computes the UNSHARDED scoring loop with the HIP kernels and
its shard of the candidate-sharded one (pipelined all-gather of score blocks; all-reduce-of-counts ranking; all-gather of best-5 lists) and asserts that the
sharded results are the unsharded ones BIT FOR BIT -- the collectives only move what the same kernels computed.  Prints "OK <rank>".
`dist_gpu_worker.py dp`: the data-parallel TRAINING step instead (dp_training_step below).  Reference for the partition:
trainer/trainer.py:52-56 (a query's 1 + negatives stay together), model/loss.py:52-57 (sum reduction: gradients add over ranks)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _init():
    """gloo with every rank on cuda:0 (a one-GPU box: the collective LOGIC over the real kernels), or -- TXE_TEST_BACKEND=nccl, a box with
    at least WORLD_SIZE GPUs -- RCCL with one GPU per rank (the real transport: async collectives on RCCL's own streams)"""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("TXE_TEST_BACKEND", "gloo")
    index = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, dev


def main():
    rank, world, dev = _init()
    from taxoexpan_amd import model_zoo as mz, ops
    from taxoexpan_amd.scoring import rank_all_fused, score_all, score_all_sharded, shard_bounds, topk_parents, topk_parents_fused
    solo = [dist.new_group([rr]) for rr in range(world)][rank]      # (every rank creates every group: new_group is collective)
    G, Q, l, r = 10007, 333, 500, 250                      # (G: no multiple of the world sizes, of 4 or of the tile widths)
    gen = torch.Generator().manual_seed(123)
    hg = (torch.randn(G, l, generator=gen) * 0.3).to(dev)
    queries = torch.nn.functional.normalize(torch.randn(Q, r, generator=gen), dim=1).to(dev)
    for kind in ("LBM", "BIM"):
        torch.manual_seed(5)
        match = getattr(mz, kind)(l, r).to(dev)
        rs = np.random.RandomState(9)
        npos = rs.randint(1, 4, size=Q)
        pos_off = np.concatenate([[0], np.cumsum(npos)])
        pos_idx = np.concatenate([rs.choice(G, size=k, replace=False) for k in npos])
        with torch.no_grad():
            S_full = score_all(match, hg, queries, block=128)
            ranks_full = rank_all_fused(match, hg, queries, pos_off, pos_idx, block=128, group=solo)     # (a 1-rank group: local)
            lo, hi = shard_bounds(G, world, rank)
            blocks = {}

            def on_block(q0, blk):
                for rr in range(world):                    # read in place, per shard, like a consumer would
                    a, b = shard_bounds(G, world, rr)
                    assert torch.equal(blk.columns(rr), S_full[q0:q0 + blk.shape[0], a:b]), (kind, q0, rr)
                blocks[q0] = blk.dense().clone()
            score_all_sharded(match, hg[lo:hi], G, queries, block=128, on_block=on_block)
            S_sh = torch.cat([blocks[k] for k in sorted(blocks)], 0)
            assert torch.equal(S_sh, S_full), kind
            S_sh2 = score_all_sharded(match, hg[lo:hi], G, queries, block=128)          # the default collecting path
            assert torch.equal(S_sh2, S_full), kind
            ranks_sh = rank_all_fused(match, hg[lo:hi], queries, pos_off, pos_idx, block=128, shard_lo=lo, sharded=True)
            assert torch.equal(ranks_sh, ranks_full), kind
            off_t, idx_t = torch.tensor(pos_off, dtype=torch.int32), torch.tensor(pos_idx, dtype=torch.int32)
            assert torch.equal(ops.rank_block(S_full, off_t, idx_t, True), ranks_full), kind
            # the 5 best parents (infer.py:96-106): every rank selects among its shard with the fused kernels, the [Q, 5] lists are
            # all-gathered and merged -- the unsharded selection, which is the stable sort of the materialised scores
            ids = torch.arange(G, device=dev)
            for larger in (True, False):
                want = topk_parents(S_full, ids, 5, larger)
                assert torch.equal(topk_parents_fused(match, hg, queries, None, 5, larger, group=solo), want), (kind, larger)
                assert torch.equal(topk_parents_fused(match, hg[lo:hi], queries, None, 5, larger, block=128, shard_lo=lo, sharded=True), want), (kind, larger)
            # evaluate() / infer() on ONE rank of the initialised world, whole candidate list, no group: no collective may be issued
            # (the peers are not calling -- a hang here is the failure), same results
            if rank == 0:
                assert torch.equal(topk_parents_fused(match, hg, queries, None, 5, True), topk_parents(S_full, ids, 5, True)), kind
                assert torch.equal(rank_all_fused(match, hg, queries, pos_off, pos_idx, block=128), ranks_full), kind
    torch.cuda.synchronize()
    dist.barrier()
    print(f"OK {rank}", flush=True)
    dist.destroy_process_group()


def _shard_batch(k, m, x, qf, q_lo, q_hi, per_query):
    """the egonets of queries [q_lo, q_hi) of a batch given by its egonet shapes k, m (grand-parents / siblings per egonet), node
    features x and stacked query rows qf -- the same egonets, renumbered from 0"""
    from taxoexpan_amd.graph import BatchedDGLGraph
    g0, g1 = q_lo * per_query, q_hi * per_query
    n = k + 1 + m
    noff = np.concatenate([[0], np.cumsum(n)])
    g = BatchedDGLGraph.from_egonet_shapes(k[g0:g1], m[g0:g1])
    return g, x[int(noff[g0]):int(noff[g1])], qf[g0:g1]


def dp_training_step():
    """Data-parallel training on the HIP kernels at world size N (all ranks on cuda:0, gloo): rank r takes the queries
    shard_bounds(n_queries, N, r) of ONE batch -- a query's 1 + 31 egonets stay together, so its InfoNCE row is local -- runs the real
    TaxoExpan step inside overlapped_gradient_allreduce(model=...) (the stack announces its layer buckets, the matcher's goes out from
    its hook) and reduces the rest with allreduce_gradients.  Every parameter gradient must equal (a) the sum of the shards' gradients
    computed in ONE process without any collective, to 2e-6 -- the collectives add nothing but the sum -- and that sum must be (b) the
    single-process step on the whole batch (the loss is a sum over queries).  Dropout off (the keep masks hash batch row indices,
    which differ between the whole batch and a shard).  Last scenario: fewer queries than ranks -- the last rank's shard is empty, it
    never runs backward, and must neither hang nor change the sums."""
    rank, world, dev = _init()
    from taxoexpan_amd import TaxoExpan, synthetic as syn
    from taxoexpan_amd.loss import info_nce_loss
    from taxoexpan_amd.scoring import allreduce_gradients, gradient_bucket_plan, overlapped_gradient_allreduce, shard_bounds
    NEG = 31
    tax = syn.make_taxonomy(6000, 9500, 250, seed=11)
    cases = [("PGAT [4,1] MAG dims", dict(in_dim=250, hidden_dim=500, out_dim=500, pos_dim=50, num_layers=1, heads=[4, 1]), 128),
             ("PGAT [2,4,1] three layers", dict(in_dim=250, hidden_dim=32, out_dim=48, pos_dim=50, num_layers=2, heads=[2, 4, 1]), 64),
             ("PGAT [4,1], last shard empty", dict(in_dim=250, hidden_dim=64, out_dim=64, pos_dim=50, num_layers=1, heads=[4, 1]), world - 1)]
    for name, dims, n_queries in cases:
        torch.manual_seed(47)
        model = TaxoExpan("PGAT", "WMR", "LBM", **dict(dims, feat_drop=0.0, attn_drop=0.0, hidden_drop=0.0, out_drop=0.0)).to(dev).train()
        with torch.no_grad():                 # (spread the scores a little; LBM's exp under InfoNCE's softmax saturates quickly -- a row
            model.match.W.weight.mul_(2.0)    #  whose positive wins outright has an exactly zero gradient)
        params = list(model.parameters())
        g, qf, _ = syn.training_batch(tax, n_queries, NEG, seed=77)
        x = g.ndata.pop("x")
        pos = g.ndata["pos"].numpy()
        nn_ = np.asarray(g.batch_num_nodes)
        noff = np.concatenate([[0], np.cumsum(nn_)])
        k = np.asarray([int((pos[noff[i]:noff[i + 1]] == 0).sum()) for i in range(len(nn_))])
        m = nn_ - 1 - k
        def local_step(q_lo, q_hi):
            for p in params:
                p.grad = None
            gs, xs, qs = _shard_batch(k, m, x, qf, q_lo, q_hi, 1 + NEG)
            pred = model(gs, xs.to(dev), qs.to(dev))
            info_nce_loss(pred.reshape(q_hi - q_lo, -1), torch.zeros(q_hi - q_lo, dtype=torch.long, device=dev)).backward()
            return [p.grad.detach().clone() for p in params]
        # ---- (1) the single-process step on the whole batch ----
        whole = local_step(0, n_queries)
        assert all(bool(torch.isfinite(w).all()) for w in whole), name
        assert max(float(w.abs().max()) for w in whole) > 1e-6, (name, [float(w.abs().max()) for w in whole])
        # ---- (2) the same partition WITHOUT collectives: every shard's step in this process, gradients added in rank order ----
        want = None
        for rr in range(world):
            a_, b_ = shard_bounds(n_queries, world, rr)
            if b_ > a_:
                gr = local_step(a_, b_)
                want = gr if want is None else [u + v for u, v in zip(want, gr)]
        # the partition is right: sum of the shards' gradients = the whole batch's.  Rows that fall into another tile round of a GEMM
        # are summed in another k order (last bit), and a leaky_relu whose pre-activation sits within that bit of 0 then takes the other
        # branch on ONE element -- which moves one weight row's gradient by a single node's term.  Hence: 99.9 % of the entries at
        # 1e-4 / 1e-5, every entry within 3 % of the tensor's largest.
        for (pn, p), w, v in zip(model.named_parameters(), whole, want):
            d, mx = (v - w).abs(), float(w.abs().max())
            frac_bad = float((d > 1e-4 * w.abs() + 1e-5 * mx + 1e-9).float().mean())
            assert frac_bad <= 1e-3 and float(d.max()) <= 3e-2 * mx + 1e-9, (name, pn, frac_bad, float(d.max()), mx)
        # ---- (3) the data-parallel step: what the collectives add must be NOTHING but the sum (same kernels, same shard batches:
        #      the per-rank gradients are bit-identical to (2)'s, only the order of the sum over ranks may differ) ----
        for p in params:
            p.grad = None
        lo, hi = shard_bounds(n_queries, world, rank)
        plan = gradient_bucket_plan(model)
        assert [l for l, _ in plan] == list(range(dims["num_layers"], 0, -1)), name
        with overlapped_gradient_allreduce(model=model) as ov:
            if hi > lo:
                gs, xs, qs = _shard_batch(k, m, x, qf, lo, hi, 1 + NEG)
                pred = model(gs, xs.to(dev), qs.to(dev))
                info_nce_loss(pred.reshape(hi - lo, -1), torch.zeros(hi - lo, dtype=torch.long, device=dev)).backward()
        planned = {id(p) for _, ps in plan for p in ps} | {id(p) for p in model.match.parameters()}
        assert ov.reduced == planned, (name, len(ov.reduced), len(planned))      # the layer buckets and the matcher's, on every rank
        allreduce_gradients(params, skip=ov)
        torch.cuda.synchronize()
        for (pn, p), w in zip(model.named_parameters(), want):
            tol = 2e-6 * w.abs() + 2e-6 * float(w.abs().max()) + 1e-12
            bad = (p.grad - w).abs() > tol
            assert not bool(bad.any()), (name, pn, int(bad.sum()), float((p.grad - w).abs().max()), float(w.abs().max()))
    dist.barrier()
    print(f"OK {rank}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    dp_training_step() if sys.argv[1:2] == ["dp"] else main()
