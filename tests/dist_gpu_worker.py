"""TEST INFRASTRUCTURE: one rank of a world-size-N job whose ranks all sit on cuda:0 (a single-GPU box) over gloo, launched by
tests/test_gpu_eval_infer.py through torch.distributed.run.  Every rank computes the UNSHARDED scoring loop with the HIP kernels and
its shard of the candidate-sharded one (pipelined all-gather of score blocks; all-reduce-of-counts ranking) and asserts that the
sharded results are the unsharded ones BIT FOR BIT -- the collectives only move what the same kernels computed.  Prints "OK <rank>"."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from taxoexpan_amd import model_zoo as mz, ops
    from taxoexpan_amd.scoring import rank_all_fused, score_all, score_all_sharded, shard_bounds
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
            ranks_sh = rank_all_fused(match, hg[lo:hi], queries, pos_off, pos_idx, block=128, shard_lo=lo)
            assert torch.equal(ranks_sh, ranks_full), kind
            off_t, idx_t = torch.tensor(pos_off, dtype=torch.int32), torch.tensor(pos_idx, dtype=torch.int32)
            assert torch.equal(ops.rank_block(S_full, off_t, idx_t, True), ranks_full), kind
    torch.cuda.synchronize()
    dist.barrier()
    print(f"OK {rank}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
