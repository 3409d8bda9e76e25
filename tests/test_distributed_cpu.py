"""CPU, world_size 2 over gloo: the multi-GPU logic (candidate sharding + score all-gather, gradient all-reduce) is
exercised with an injected CPU local-score function (test infrastructure only -- the product default is the HIP kernel)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, G, Q, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from taxoexpan_amd import scoring
        rs = np.random.RandomState(0)
        U = torch.from_numpy(rs.standard_normal((G, 6)).astype(np.float32))
        Qm = torch.from_numpy(rs.standard_normal((Q, 6)).astype(np.float32))
        lo, hi = scoring.shard_bounds(G, world, rank)
        U_loc = U[lo:hi]

        def local_fn(qb, out):
            out[:, :U_loc.shape[0]] = qb @ U_loc.t()
        S = scoring.score_all_sharded(None, U_loc, G, Qm, block=3, local_score_fn=local_fn)
        full = Qm @ U.t()
        ok1 = torch.allclose(S, full, atol=1e-6) and S.shape == (Q, G)
        # gradient all-reduce: each rank holds different grads, sum must be identical everywhere
        ps = [torch.nn.Parameter(torch.zeros(3, 2)), torch.nn.Parameter(torch.zeros(5))]
        ps[0].grad = torch.full((3, 2), float(rank + 1))
        ps[1].grad = torch.arange(5.0) * (rank + 1)
        scoring.allreduce_gradients(ps)
        tot = sum(range(1, world + 1))
        ok2 = torch.equal(ps[0].grad, torch.full((3, 2), float(tot))) and torch.equal(ps[1].grad, torch.arange(5.0) * tot)
        ret[rank] = bool(ok1 and ok2)
    finally:
        dist.destroy_process_group()


def _run(G, Q, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, G, Q, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_sharded_scoring_and_grad_allreduce_world2_even():
    _run(G=10, Q=7)


def test_sharded_scoring_world2_ragged_and_tiny():
    _run(G=11, Q=4)     # shards 6 + 5: padding only at the very end
    _run(G=1, Q=2)      # second rank owns nothing


def test_shard_bounds_cover_exactly():
    from taxoexpan_amd.scoring import shard_bounds
    for n in (0, 1, 7, 8, 24754, 355808):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            c = -(-n // w) if n else 0
            assert all(hi - lo <= c for lo, hi in spans)


def _fused_rank_worker(rank, world, port, q):
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taxoexpan_amd import scoring
    rs = np.random.RandomState(0)
    G, nq, r = 203, 37, 6
    U = torch.from_numpy(np.round(rs.randn(G, r) * 2).astype(np.float32))       # integer-valued: exact scores, many ties
    Qm = torch.from_numpy(np.round(rs.randn(nq, r) * 2).astype(np.float32))
    npos = rs.randint(1, 4, size=nq)
    pos_idx = np.concatenate([rs.choice(G, size=k, replace=False) for k in npos]).astype(np.int64)
    pos_off = np.concatenate([[0], np.cumsum(npos)]).astype(np.int64)
    lo, hi = scoring.shard_bounds(G, world, rank)
    Ul = U[lo:hi]

    def f_thr(qb, off, idx_local):
        cnt = (off[1:] - off[:-1]).long()
        qid = torch.repeat_interleave(torch.arange(qb.shape[0]), cnt)
        ok = idx_local >= 0
        s = (qb[qid] * Ul[idx_local.clamp(min=0).long()]).sum(1)
        return torch.where(ok, s, torch.zeros_like(s))

    def f_cnt(qb, off, thr):
        S = qb @ Ul.t()
        cnt = (off[1:] - off[:-1]).long()
        qid = torch.repeat_interleave(torch.arange(qb.shape[0]), cnt)
        return (S[qid] > thr[:, None]).sum(1).to(torch.int32)

    got = scoring.rank_all_fused(None, Ul, Qm, pos_off, pos_idx, block=16, shard_lo=lo, local_fns=(f_thr, f_cnt))
    S = (Qm @ U.t()).numpy()
    want = []
    for i in range(nq):
        P = pos_idx[pos_off[i]:pos_off[i + 1]]
        neg = np.ones(G, dtype=bool)
        neg[P] = False
        want += [1 + int((S[i][neg] > S[i][p]).sum()) for p in P]
    q.put((rank, got.tolist() == want))
    dist.destroy_process_group()


def test_fused_rank_counts_all_reduce_world2():
    """candidate-sharded fused ranking: thresholds and counts all-reduced over 2 gloo ranks == single-process metric ranks"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650
    procs = [ctx.Process(target=_fused_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res)
