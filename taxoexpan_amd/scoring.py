"""The batched egonet scoring loop of test_fast.py:99-225 / infer.py:82-159, MI355X-native.

Reference: encode every candidate egonet once, then for each query expand it to G rows and call model.match -- a
G x l x r bilinear per query, a D2H copy per query, numpy ranking per query.
Here:  U = HG W is formed once (txe_bilinear_project); a block of queries is ONE fp32-MFMA GEMM with the exp fused
(txe_score_block); ranks are counted on device (txe_rank_block).  Multi-GPU: candidates are sharded contiguously over
the ranks of one node; each rank scores its shard and the score blocks are all-gathered over xGMI (RCCL) so that every
rank holds the full [queries x candidates] block, as the north star asks.
"""
import math

import torch
import torch.distributed as dist

from . import ops


def score_all(match, hg, queries, block=1024, out=None):
    """S[q][g] = match(hg[g], queries[q]) for all pairs; match is a BIM / LBM module."""
    U = ops.bilinear_project(hg, match.W.weight)
    Q = queries.shape[0]
    G = hg.shape[0]
    S = out if out is not None else torch.empty((Q, (G + 3) // 4 * 4), dtype=torch.float32, device=hg.device)[:, :G]
    for q0 in range(0, Q, block):
        ops.score_block(queries[q0:q0 + block], U, match.apply_exp, out=S[q0:q0 + block])
    return S


def encode_candidates(model, graph, chunk=None, device=None):
    """encode_graph over all candidate egonets (test_fast.py:99-108 small mode; :149-179 chunks of `-b` egonets).
    `graph` is one batched graph (chunk=None) or a list of batched graphs; features are taken from ndata['x']."""
    graphs = graph if isinstance(graph, (list, tuple)) else [graph]
    device = device or next(model.parameters()).device
    outs = []
    with torch.no_grad():
        for bg in graphs:
            h = bg.ndata['x'].to(device, non_blocking=True)
            pos = bg.ndata['pos'].to(device)
            had_pos = 'pos' in bg.ndata
            bg.ndata['h'] = model.graph_propagate(bg, h)
            outs.append(model.readout(bg, pos))
            if had_pos:
                bg.ndata['pos'] = pos          # PGAT/PGCN pop it (model_zoo.py:163,212); keep the graph reusable
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


# ---------------------------------------------------------------------------------------------------------------
# sharding helpers (pure index math: unit-tested on CPU)
# ---------------------------------------------------------------------------------------------------------------
def shard_bounds(n, world, rank):
    """contiguous shard [lo, hi) of n items: every rank but the last gets ceil(n/world) items, so the concatenation of
    equal-width padded shards has its padding only at the very end."""
    c = math.ceil(n / world) if world > 0 else n
    lo = min(rank * c, n)
    return lo, min(lo + c, n)


def all_gather_score_block(local_block, n_total, group=None):
    """local_block [nq, c] (this rank's candidate shard, padded to the common width c = ceil(G/world)) ->
    full [nq, G] on every rank.  One RCCL all-gather; the only data-path collective of inference."""
    world = dist.get_world_size(group)
    nq, c = local_block.shape
    buf = torch.empty((world * nq, c), dtype=local_block.dtype, device=local_block.device)   # rank-major concatenation
    dist.all_gather_into_tensor(buf, local_block.contiguous(), group=group)
    return buf.view(world, nq, c).permute(1, 0, 2).reshape(nq, world * c)[:, :n_total]


def score_all_sharded(match, hg_local, n_total, queries, block=1024, group=None, local_score_fn=None, on_block=None):
    """Candidate-sharded scoring.  hg_local: this rank's [hi-lo, l] candidate representations (shard_bounds order).
    For every query block: local scores -> all-gather -> on_block(q0, S_full [nq, G]) (default: collect and return).
    local_score_fn(queries_block, out_padded) is injectable so the collective logic is testable without a GPU."""
    world = dist.get_world_size(group)
    c = math.ceil(n_total / world)
    dev = hg_local.device
    if local_score_fn is None:
        U = ops.bilinear_project(hg_local, match.W.weight) if hg_local.shape[0] > 0 else hg_local.new_zeros((0, queries.shape[1]))

        def local_score_fn(qb, out):
            if U.shape[0] > 0:
                ops.score_block(qb, U, match.apply_exp, out=out[:, :U.shape[0]])
    collected = []
    for q0 in range(0, queries.shape[0], block):
        qb = queries[q0:q0 + block]
        loc = torch.zeros((qb.shape[0], c), dtype=torch.float32, device=dev)
        local_score_fn(qb, loc)
        full = all_gather_score_block(loc, n_total, group)
        if on_block is not None:
            on_block(q0, full)
        else:
            collected.append(full)
    return None if on_block is not None else torch.cat(collected, 0)


def allreduce_gradients(params, group=None):
    """Data-parallel training: the InfoNCE loss is a SUM over queries (loss.py:57) and queries are sharded over ranks,
    so gradients simply add: one flat bucket (1.76 M fp32 = 7 MB for the MAG config), one RCCL all-reduce."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    parts = flat.split([g.numel() for g in grads])
    torch._foreach_copy_(grads, [q.view_as(g) for q, g in zip(parts, grads)])      # one multi-tensor kernel


def topk_parents(S, candidate_ids, k=5, larger_is_better=True):
    """infer.py:100-106: the k best candidate positions per query (descending score for info_nce, else ascending)."""
    idx = torch.topk(S, k=min(k, S.shape[1]), dim=1, largest=larger_is_better).indices
    return candidate_ids.to(idx.device)[idx]
