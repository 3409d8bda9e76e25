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


def score_all(match, hg, queries, block=None, out=None):
    """S[q][g] = match(hg[g], queries[q]) for all pairs; match is a BIM / LBM module.
    block: queries per GEMM launch; default: a query set of up to 4,096 goes in ONE launch (MAG-CS: 2,459 queries x 24.7 k candidates
    are 7.6 rounds of tiles -- in blocks of 1,024 every block pays its own partial last round and a second, 16-tile launch), larger
    sets in blocks of 1,024."""
    U = ops.bilinear_project(hg, match.W.weight)
    Q = queries.shape[0]
    G = hg.shape[0]
    if block is None:
        block = 1024 if Q > 4096 else max(int(Q), 1)
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
    total = sum(int(bg.number_of_nodes()) for bg in graphs)
    with torch.no_grad(), ops.projection_cache(total):   # weights are fixed for the pass: table projections are shared by the chunks
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


class ShardedScoreBlock:
    """One query block of the candidate-sharded scoring loop after its all-gather: `shards` [world, nq, c] holds, for rank r, the
    scores of candidates [r*c, (r+1)*c) (c = ceil(G / world); the tail of the last shard is padding).  Consumers index it in place --
    `columns(r)` is rank r's [nq, n_r] slab -- or ask for the dense [nq, G] matrix, which costs one more pass over the block."""

    def __init__(self, shards, n_total):
        self.shards, self.n_total = shards, int(n_total)

    @property
    def shape(self):
        return (self.shards.shape[1], self.n_total)

    def columns(self, r):
        c = self.shards.shape[2]
        return self.shards[r, :, :max(0, min(c, self.n_total - r * c))]

    def dense(self):
        world, nq, c = self.shards.shape
        return self.shards.permute(1, 0, 2).reshape(nq, world * c)[:, :self.n_total]


def all_gather_score_block(local_block, n_total, group=None):
    """local_block [nq, c] (this rank's candidate shard, padded to the common width c = ceil(G/world)) ->
    full [nq, G] on every rank.  One RCCL all-gather (blocking form; score_all_sharded pipelines it)."""
    world = dist.get_world_size(group)
    nq, c = local_block.shape
    buf = torch.empty((world, nq, c), dtype=local_block.dtype, device=local_block.device)   # rank-major concatenation
    dist.all_gather_into_tensor(buf.view(world * nq, c), local_block.contiguous(), group=group)
    return ShardedScoreBlock(buf, n_total).dense()


def score_all_sharded(match, hg_local, n_total, queries, block=1024, group=None, local_score_fn=None, on_block=None, pipeline=True):
    """Candidate-sharded scoring.  hg_local: this rank's [hi-lo, l] candidate representations (shard_bounds order).
    For every query block: local scores -> all-gather over xGMI -> on_block(q0, ShardedScoreBlock) (default: collect the dense
    [Q, G] matrix and return it).
    Pipelined (a 1,024-query MAG-Full block is 182 MB per rank, ~8 ms on the ring against ~0.2 ms of GEMM): the gather of
    block i is issued asynchronously (RCCL's own stream) and waited for only after block i+1's local GEMM has been enqueued, on
    THREE rotating pairs of preallocated buffers -- nothing is allocated, zero-filled or permuted inside the loop, and the consumer
    reads the [world, nq, c] gather buffer in place.  A block handed to on_block stays valid until on_block is called again (the
    gather issued in between writes the third buffer; with two, block i-1's buffer was re-used by gather i+1 BEFORE on_block(i) ran).
    local_score_fn(queries_block, out_padded) is injectable so the collective logic is testable without a GPU."""
    world = dist.get_world_size(group)
    c = math.ceil(n_total / world)
    dev = hg_local.device
    if local_score_fn is None:
        U = ops.bilinear_project(hg_local, match.W.weight) if hg_local.shape[0] > 0 else hg_local.new_zeros((0, queries.shape[1]))

        def local_score_fn(qb, out):
            if U.shape[0] > 0:
                ops.score_block(qb, U, match.apply_exp, out=out[:, :U.shape[0]])
    Q = queries.shape[0]
    bmax = min(block, max(Q, 1))
    nbuf = 3 if pipeline else 1
    loc = [torch.zeros((bmax, c), dtype=torch.float32, device=dev) for _ in range(nbuf)]     # padding columns: zeroed once, never written
    full = [torch.empty((world * bmax * c,), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    collected = []

    def consume(q0, nq, b, work):
        if work is not None:
            work.wait()                                   # the CURRENT stream waits for the collective; the host does not block
        blk = ShardedScoreBlock(full[b][:world * nq * c].view(world, nq, c), n_total)
        if on_block is not None:
            on_block(q0, blk)
        else:
            d = blk.dense()                               # (world > 1: a fresh tensor already; world == 1: a view of the gather buffer)
            collected.append(d.clone() if (pipeline and world == 1) else d)
    pending = None
    for i, q0 in enumerate(range(0, Q, block)):
        qb = queries[q0:q0 + block]
        nq, b = qb.shape[0], i % nbuf
        local_score_fn(qb, loc[b][:nq])
        out = full[b][:world * nq * c].view(world * nq, c)
        if pipeline:
            work = dist.all_gather_into_tensor(out, loc[b][:nq], group=group, async_op=True)
            if pending is not None:
                consume(*pending)                         # block i-1: its gather ran under this block's local GEMM
            pending = (q0, nq, b, work)
        else:
            dist.all_gather_into_tensor(out, loc[b][:nq], group=group)
            consume(q0, nq, b, None)
    if pending is not None:
        consume(*pending)
    return None if on_block is not None else (torch.cat(collected, 0) if collected else torch.zeros((0, n_total), device=dev))


def _sharded_call(group, sharded):
    """is this a candidate-SHARDED call (collectives inside)?  Only when the caller says so -- a process group handed over, or
    sharded=True for the default group.  An initialised world alone decides nothing: evaluate() / infer() on rank 0 of a DDP job, or on
    every rank with the whole candidate list, stay local and issue no collective."""
    on = bool(sharded) or group is not None
    if on and not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("candidate-sharded scoring asked for (group / sharded=True) but torch.distributed is not initialised")
    return on


def rank_all_fused(match, hg, queries, pos_off, pos_idx, block=None, larger_is_better=True, group=None, shard_lo=0, local_fns=None,
                   sharded=False):
    """Ranks of every query's true parents among ALL candidates without materialising the score matrix.
    hg: this rank's candidate representations (rows [shard_lo, shard_lo + len) of the global candidate list; the whole list when
    not distributed).  pos_off [Q+1] / pos_idx: GLOBAL candidate columns of each query's true parents.
    Per query block: thresholds = scores of the positives (each computed by the rank that owns the candidate, summed over ranks),
    counts = fused score-and-compare over the local shard (summed over ranks), ranks = 1 + counts - own positives that beat it.
    The only collectives are two all-reduces of [n_positives] vectors per block -- instead of the [queries x candidates] all-gather.
    Sharded ONLY when asked: `group` given, or sharded=True (the default group) -- see _sharded_call.
    local_fns = (positive_scores, score_count) is injectable so that the collective logic is testable without a GPU."""
    dev = hg.device
    distributed = _sharded_call(group, sharded)
    if block is None:
        # query blocks of 1,024 (a [1024 x G] tile pass per block; the thresholds' small score matrix grows with block x positives);
        # a query set of a few thousand goes in ONE block: five launches in all instead of five per 1,024 queries
        block = 1024 if queries.shape[0] > 4096 else max(int(queries.shape[0]), 1)
    if local_fns is None and hg.shape[0] > 0:
        return _rank_all_fused_device(match, hg, queries, pos_off, pos_idx, block, larger_is_better, group, shard_lo, distributed)
    if local_fns is None:
        U = ops.bilinear_project(hg, match.W.weight) if hg.shape[0] > 0 else None
        exp = match.apply_exp

        def f_thr(qb, off, idx_local):
            return ops.positive_scores(qb, U, exp, off, idx_local) if U is not None else torch.zeros(idx_local.numel(), device=dev)

        def f_cnt(qb, off, thr):
            if U is None:
                return torch.zeros(max(int(thr.numel()), 1), dtype=torch.int32, device=dev)
            return ops.score_count_block(qb, U, exp, off, thr, larger_is_better)
    else:
        f_thr, f_cnt = local_fns
    pos_off_h = torch.as_tensor(pos_off).to(torch.int64).cpu()            # block boundaries are host integers
    pos_off_d = pos_off_h.to(dev)                                          # one upload for the whole loop
    idx_all = torch.as_tensor(pos_idx).to(torch.int64).to(dev) - shard_lo
    n_local = hg.shape[0]
    idx_all = torch.where((idx_all >= 0) & (idx_all < n_local), idx_all, torch.full_like(idx_all, -1)).to(torch.int32)
    out = []
    if idx_all.numel() == 0:                                       # (the same early exit and the same skipped blocks as the device
        return torch.zeros(0, dtype=torch.int32, device=dev)       #  path: a rank with an empty shard issues the same collectives)
    for q0 in range(0, queries.shape[0], block):
        q1 = min(q0 + block, queries.shape[0])
        lo, hi = int(pos_off_h[q0]), int(pos_off_h[q1])
        if hi == lo:
            continue
        off = (pos_off_d[q0:q1 + 1] - lo).to(torch.int32)
        idx_local = idx_all[lo:hi]
        qb = queries[q0:q1]
        thr = f_thr(qb, off, idx_local)
        if distributed:
            dist.all_reduce(thr, op=dist.ReduceOp.SUM, group=group)          # each positive lives in exactly one shard
        counts = f_cnt(qb, off, thr)[:hi - lo]
        if distributed:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        if local_fns is None:
            out.append(ops.rank_finalize(off, thr, counts, larger_is_better))
        else:
            out.append(_rank_finalize_host(off, thr, counts, larger_is_better))
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.int32, device=dev)


def _rank_all_fused_device(match, hg, queries, pos_off, pos_idx, block, larger_is_better, group, shard_lo, distributed):
    """rank_all_fused on the HIP kernels with everything that does not depend on the scores prepared ONCE, on the host, for all query
    blocks (the positives' rows and block-local offsets): the
    loop itself is four launches per block, no host synchronisation, no per-block index arithmetic on the device.  (The first version
    did that arithmetic per block with a dozen torch launches and a repeat_interleave sync: 0.65 ms of host time against 0.69 ms of
    kernels on the MAG-CS shape -- the fused route lost to materialise + rank by 2x for that reason alone.)"""
    import numpy as np
    dev = hg.device
    n_local, Q = hg.shape[0], queries.shape[0]
    off_h = np.asarray(torch.as_tensor(pos_off).cpu(), dtype=np.int64)
    idx_h = np.asarray(torch.as_tensor(pos_idx).cpu(), dtype=np.int64) - int(shard_lo)
    n_pos = int(idx_h.shape[0])
    if n_pos == 0 or Q == 0:
        return torch.zeros(0, dtype=torch.int32, device=dev)
    nblk = (Q + block - 1) // block
    q0s = np.minimum(np.arange(nblk + 1) * block, Q)
    lo_b = off_h[q0s]                                                     # first positive of every block (+ the end)
    local_h = (idx_h >= 0) & (idx_h < n_local)
    offs_h = np.concatenate([off_h[q0s[b]:q0s[b + 1] + 1] - lo_b[b] for b in range(nblk)]).astype(np.int32)   # block-local offsets, back to back
    up = lambda a, dt: torch.as_tensor(a, dtype=dt).to(dev, non_blocking=True)
    idxc = up(np.where(local_h, idx_h, 0), torch.int64)
    offs = up(offs_h, torch.int32)
    all_local = bool(local_h.all())
    localb = None if all_local else up(local_h, torch.bool)
    U = ops.bilinear_project(hg, match.W.weight)
    exp = match.apply_exp
    Qp = ops.pad_queries_like(queries, U)
    counts = torch.zeros(n_pos, dtype=torch.int32, device=dev)
    ranks = torch.empty(n_pos, dtype=torch.int32, device=dev)
    thr_all = torch.empty(n_pos, dtype=torch.float32, device=dev)
    o0 = 0
    for b in range(nblk):
        q0, q1, lo, hi = int(q0s[b]), int(q0s[b + 1]), int(lo_b[b]), int(lo_b[b + 1])
        off = offs[o0:o0 + (q1 - q0) + 1]
        o0 += (q1 - q0) + 1
        if hi == lo:
            continue
        qb = Qp[q0:q1]
        # thresholds: the positives' scores through the SAME score kernel as the block (bit-identical values), the staircase of tiles
        # that holds them only
        thr = ops.positive_scores_staircase(qb, ops.gather_padded_rows(U, idxc[lo:hi]), exp, off, thr_all[lo:hi])
        if localb is not None:                                           # a positive that lives in another shard contributes 0 here
            thr.copy_(torch.where(localb[lo:hi], thr, torch.zeros((), device=dev)))      # (not a product: the placeholder may be inf)
        if distributed:
            dist.all_reduce(thr, op=dist.ReduceOp.SUM, group=group)      # each positive lives in exactly one shard
        ops.score_count_block(qb, U, exp, off, thr, larger_is_better, counts=counts[lo:hi], q_padded=True)
        if distributed:
            dist.all_reduce(counts[lo:hi], op=dist.ReduceOp.SUM, group=group)
        ops.rank_finalize(off, thr, counts[lo:hi], larger_is_better, out=ranks[lo:hi])
    return ranks


def _rank_finalize_host(off, thr, counts, larger_is_better):
    off, thr, counts = off.cpu().tolist(), thr.cpu(), counts.cpu().to(torch.int64)
    ranks = torch.empty(len(thr), dtype=torch.int32)
    for q in range(len(off) - 1):
        for j in range(off[q], off[q + 1]):
            t = thr[off[q]:off[q + 1]]
            ranks[j] = 1 + int(counts[j]) - int(((t > thr[j]) if larger_is_better else (t < thr[j])).sum())
    return ranks


def allreduce_gradients(params, group=None, skip=None):
    """Data-parallel training: the InfoNCE loss is a SUM over queries (loss.py:57) and queries are sharded over ranks,
    so gradients simply add: one flat bucket (1.76 M fp32 = 7 MB for the MAG config), one RCCL all-reduce.
    skip: an overlapped_gradient_allreduce whose gradients were already reduced during backward."""
    done = skip.reduced if skip is not None else ()
    todo = [p for p in params if p.requires_grad and id(p) not in done]
    if not todo:
        return
    # the bucket has the same layout on every rank whatever each rank's backward produced: a parameter without a gradient here (an
    # empty shard, a branch this rank did not take) contributes zeros and receives the sum
    for p in todo:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in todo]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    parts = flat.split([g.numel() for g in grads])
    torch._foreach_copy_(grads, [q.view_as(g) for q, g in zip(parts, grads)])      # one multi-tensor kernel


def gradient_bucket_plan(model, min_layer=1):
    """the buckets overlapped_gradient_allreduce will see, from the model alone: for every GAT layer l >= min_layer of
    `model.graph_propagate` its fc.weight / attn_l / attn_r / position-embedding table -- [(layer, [parameters])], top layer first, as
    backward announces them.  Exactly the four tensors EVERY route of ops.GATStackFunction.backward announces for a layer: the weighted
    readout's position weights (three floats, announced only by the folded route) are left to the flat bucket, so the plan holds whether
    the output layer runs folded or not.  Every rank derives the same plan, so a rank whose backward never reaches the stack can still
    issue matching collectives."""
    prop = getattr(model, "graph_propagate", None)
    layers = getattr(prop, "gat_layers", None)
    if layers is None:
        return []
    emb = getattr(prop, "prop_position_embeddings", None)
    plan = []
    L = len(layers)
    for l in range(L - 1, min_layer - 1, -1):
        ps = [layers[l].fc.weight, layers[l].attn_l, layers[l].attn_r]
        if emb is not None:
            ps.append(emb[l].weight)
        plan.append((l, ps))
    return plan


class overlapped_gradient_allreduce:
    """`with overlapped_gradient_allreduce(group, model=model) as ov: loss.backward()` -- the propagation stack's backward announces each
    layer's parameter gradients the moment they exist (ops._GRAD_READY, with the ids of their parameters); they go into one flat bucket
    per layer whose all-reduce is issued asynchronously right there, under the backward of the layers below (the output layer's 4 MB
    bucket rides under ~0.6 ms of layer-0 kernels on the MAG step), and is waited for -- by the stream, not the host -- before the
    stack hands its gradients to autograd.  `allreduce_gradients(params, skip=ov)` then reduces what is left (the first layer,
    readout, matcher).
    The collective schedule is RANK-INVARIANT when `model` is given: the buckets are planned from the model (gradient_bucket_plan) and
    go out in plan order; an announcement for a layer the plan does not hold is left to the flat bucket; a planned bucket this rank's
    backward never announced (an empty shard whose readout returned zeros without running the stack, a model that runs its layers one by
    one) is issued ON EXIT -- when .grad is final -- with what the rank holds (zeros if nothing): it receives the sum, like its peers.
    A planned layer announced out of plan order, or with other tensors than planned, cannot be reconciled mid-backward (its real gradients
    would reach autograd un-reduced while the peers' collectives go out of step) and raises.  Without `model` every rank must take the
    same route through backward.
    Not re-entrant and not thread-safe: the hooks are process-global (ops._GRAD_READY / _GRAD_FLUSH); one backward at a time."""

    def __init__(self, group=None, min_layer=1, model=None):
        self.group, self.min_layer = group, min_layer
        self.reduced, self._pending = set(), []
        self.plan = gradient_bucket_plan(model, min_layer) if model is not None else None
        self._fired = set()
        # the matcher's parameters (model.py:86) get their gradients FIRST in backward: their bucket goes out from a
        # post-accumulate hook and rides under the whole encoder backward (with the plan: rank-invariant, like the layer buckets)
        match = getattr(model, "match", None) if model is not None else None
        self._early = [p for p in match.parameters() if p.requires_grad] if match is not None else []
        self._early_done, self._hooks = False, []

    def __enter__(self):
        self._prev = (ops._GRAD_READY, ops._GRAD_FLUSH)
        ops._GRAD_READY, ops._GRAD_FLUSH = self._ready, self._flush
        if self._early:
            left = {id(p) for p in self._early}

            def hook(p):
                left.discard(id(p))
                if not left and not self._early_done:        # the last of the matcher's gradients has been accumulated
                    self._reduce_early()
            self._hooks = [p.register_post_accumulate_grad_hook(hook) for p in self._early]
        return self

    def _reduce_early(self):
        self._early_done = True
        grads = [(p.grad if p.grad is not None else torch.zeros_like(p)) for p in self._early]
        for p, g in zip(self._early, grads):
            if p.grad is None:
                p.grad = g
        flat = torch.cat([g.reshape(-1) for g in grads])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, flat, grads, [id(p) for p in self._early]))

    def __exit__(self, *exc):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._early and not self._early_done and exc[0] is None:
            self._reduce_early()                             # (a rank without a loss term: zeros, FIRST like on its peers)
        self._flush()
        ops._GRAD_READY, ops._GRAD_FLUSH = self._prev
        if self.plan is not None and exc[0] is None:
            self._issue_missing()
        return False

    def _ready(self, layer, tensors, param_ids):
        if self.plan is not None:
            want = next((ps for l, ps in self.plan if l == layer), None)
            have = {i: t for t, i in zip(tensors, param_ids) if t is not None}
            if want is None or layer in self._fired:
                return                                     # not a planned bucket: the flat bucket after backward takes it
            if any(id(p) not in have or have[id(p)].numel() != p.numel() for p in want):
                raise RuntimeError(f"overlapped_gradient_allreduce: layer {layer} announced other gradients than gradient_bucket_plan() "
                                   "holds for it -- the ranks' collectives would go out of step")
            skipped = [l for l, _ in self.plan if l > layer and l not in self._fired]
            if skipped:                                    # (buckets fire top layer first, on every rank)
                raise RuntimeError(f"overlapped_gradient_allreduce: layer {layer} announced before the planned layers {skipped} above it")
            self._fired.add(layer)
            tensors, ids = [have[id(p)] for p in want], [id(p) for p in want]
        else:
            keep = [(t, i) for t, i in zip(tensors, param_ids) if t is not None]
            if layer < self.min_layer or not keep:           # the bottom layer's gradients arrive last: nothing left to hide them under
                return
            tensors, ids = [t for t, _ in keep], [i for _, i in keep]
        flat = torch.cat([t.reshape(-1) for t in tensors])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, flat, tensors, ids))

    def _issue_missing(self):
        """on exit (every .grad is final): the planned buckets this rank's backward never announced, in plan order -- all-reduce whatever
        gradient the rank holds for them (zeros if none), keep the sum"""
        for l, ps in self.plan:
            if l in self._fired:
                continue
            self._fired.add(l)
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])   # (what this rank has)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            for p, q in zip(ps, flat.split([p.numel() for p in ps])):
                p.grad = q.view_as(p).clone() if p.grad is None else p.grad.copy_(q.view_as(p))
            self.reduced.update(id(p) for p in ps)

    def _flush(self):
        for work, flat, tensors, ids in self._pending:
            work.wait()
            torch._foreach_copy_(tensors, [q.view_as(t) for q, t in zip(flat.split([t.numel() for t in tensors]), tensors)])
            self.reduced.update(ids)
        self._pending = []


def topk_parents_fused(match, hg, queries, candidate_ids=None, k=5, larger_is_better=True, block=None, group=None, shard_lo=0, sharded=False):
    """infer.py:96-106 / test_fast.py:121-131 without the score matrix: per query block ONE launch of the score GEMM whose epilogue keeps
    each tile's best k columns per row (txe_score_topk_block) + a merge launch -- no [Q, G] scores, no [Q, G] index temporaries (the
    torch composite topk_parents below needs two int64 [Q, G] ones: 3.5 GB per 1,024-query block on MAG-Full).  Same selection and
    order as topk_parents on the materialised scores of the same kernel (bit-identical values): better score first, ties by ascending
    candidate position (Python's stable sort), NaN last.  match: BIM / LBM.  hg: this rank's candidate rows (positions
    [shard_lo, shard_lo + len) of the global list).  Candidate-sharded (ONLY when asked: `group` given or sharded=True, _sharded_call
    -- every rank then holds a DISJOINT slice, which the merge relies on): every rank's [Q, k] lists
    are all-gathered (k * 8 bytes per query instead of the [queries x candidates] block) and merged by the same kernel.
    Returns candidate_ids[...] (or the positions themselves when candidate_ids is None) [Q, min(k, G)]."""
    dev = hg.device
    distributed = _sharded_call(group, sharded)
    Q, G = queries.shape[0], hg.shape[0]
    k_loc = min(int(k), G, 8)
    assert int(k) <= 8, "topk_parents_fused: k <= 8 (the kernels keep 8 entries per list)"
    if block is None:
        block = 1024 if Q > 4096 else max(Q, 1)
    outs = []
    if k_loc > 0 and Q > 0:
        U = ops.bilinear_project(hg, match.W.weight)
        Qp = ops.pad_queries_like(queries, U)
        scratch = {}
        for q0 in range(0, Q, block):
            outs.append(ops.score_topk_block(Qp[q0:q0 + block], U, match.apply_exp, k_loc, larger_is_better, idx_base=shard_lo, q_padded=True,
                                             scratch=scratch))
    if outs:
        idx, key = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    else:
        idx = torch.full((Q, 0), 0, dtype=torch.int32, device=dev)
        key = torch.zeros((Q, 0), dtype=torch.float32, device=dev)
    if distributed:
        world = dist.get_world_size(group)
        kk = min(int(k), 8)
        # every rank contributes kk slots per query (a rank with fewer candidates pads with empty slots)
        pad_i = torch.full((Q, kk), 0x7fffffff, dtype=torch.int32, device=dev)
        pad_k = torch.full((Q, kk), -float("inf"), dtype=torch.float32, device=dev)
        pad_i[:, :idx.shape[1]] = torch.where(idx >= 0, idx, torch.full_like(idx, 0x7fffffff))
        pad_k[:, :key.shape[1]] = key
        all_i = torch.empty((world, Q, kk), dtype=torch.int32, device=dev)
        all_k = torch.empty((world, Q, kk), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(all_i.view(world * Q, kk), pad_i, group=group)
        dist.all_gather_into_tensor(all_k.view(world * Q, kk), pad_k, group=group)
        cat_i = all_i.permute(1, 0, 2).reshape(Q, world * kk).contiguous()
        cat_k = all_k.permute(1, 0, 2).reshape(Q, world * kk).contiguous()
        n_real = int((cat_i[0] != 0x7fffffff).sum().item()) if Q > 0 else 0      # = min(total candidates, world * kk): the same for every query
        idx, key = ops.topk_merge(cat_k, cat_i, kk) if Q > 0 else (cat_i, cat_k)
        idx = idx[:, :min(kk, n_real)]
    if candidate_ids is None:
        return idx.long()
    return candidate_ids.to(dev)[idx.long()]


def topk_parents(S, candidate_ids, k=5, larger_is_better=True):
    """infer.py:100-106 / test_fast.py:125-131: the k best candidate positions per query -- `sorted(enumerate(scores), key=-score)[:k]`
    (descending score for info_nce, ascending otherwise).  Python's sort is stable, so equal scores come out in ascending candidate
    order, and that is what this returns (torch.topk alone leaves the order of ties unspecified): the strictly better candidates,
    then the lowest-index members of the tie group at the k-th value.  S [Q, G] -> candidate_ids[...] [Q, min(k, G)]."""
    Q, G = S.shape
    k = min(int(k), G)
    if k == 0 or Q == 0:
        return candidate_ids.new_zeros((Q, 0)).to(S.device)
    key = S if larger_is_better else -S
    # a NaN score compares false with everything: Python's sorted() leaves it wherever it happens to stand, here it ranks last (and the
    # selection below never indexes past the row: without this a row holding a NaN selected only filler columns)
    key = torch.where(torch.isnan(key), torch.full_like(key, -float("inf")), key)
    vk = torch.topk(key, k, dim=1).values[:, -1:]                                     # the k-th best value of each row
    ar = torch.arange(G, device=S.device).expand(Q, G)
    fill = torch.full_like(ar, G)
    better = torch.topk(torch.where(key > vk, ar, fill), k, dim=1, largest=False).values     # <= k-1 real columns, ascending, then G
    tied = torch.topk(torch.where(key == vk, ar, fill), k, dim=1, largest=False).values      # lowest k columns of the tie group
    cand = torch.cat([better, tied], 1)                                                       # [Q, 2k], fillers = G
    ckey = torch.where(cand < G, torch.gather(key, 1, cand.clamp(max=G - 1)), torch.full((), -float("inf"), device=S.device, dtype=key.dtype))
    # order by (key descending, column ascending): columns are ascending inside each half; one stable sort by column, one by key
    o1 = torch.sort(cand, dim=1, stable=True).indices
    cand, ckey = torch.gather(cand, 1, o1), torch.gather(ckey, 1, o1)
    o2 = torch.sort(ckey, dim=1, descending=True, stable=True).indices
    idx = torch.gather(cand, 1, o2)[:, :k].clamp(max=G - 1)
    return candidate_ids.to(idx.device)[idx]
