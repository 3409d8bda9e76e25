#!/usr/bin/env python3
"""The training step at shapes OTHER than the BASELINE configs -- model families, widths, head counts, position widths -- with the top
kernels' durations and roofline fractions (libtxe's own HIP-event profiler, as bench.py's roofline block): the hunt for kernels that fall
off their fast path at a shape nobody tuned (round 5 found two this way: the fused backward sweep spilling at 2,400-column rows, the Z
sweep at rows of 4 / 8 / 16 column tiles).      gpurun -- python tools/shape_sweep.py [top-N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from taxoexpan_amd import TaxoExpan, synthetic as syn
from taxoexpan_amd.optim import Adam
dev = torch.device("cuda:0")
TOP = int(sys.argv[1]) if len(sys.argv) > 1 else 7
torch.autograd.set_multithreading_enabled(False)
CASES = [
    ("PGAT MAG dims (BASELINE configs[1])", dict(prop="PGAT", in_dim=250, hidden_dim=500, out_dim=500, pos_dim=50, heads=[4, 1])),
    ("GAT no-pos MAG dims", dict(prop="GAT", in_dim=250, hidden_dim=500, out_dim=500, pos_dim=50, heads=[4, 1])),
    ("PGAT pos 16 (8 column tiles)", dict(prop="PGAT", in_dim=250, hidden_dim=500, out_dim=500, pos_dim=16, heads=[4, 1])),
    ("PGAT SemEval dims", dict(prop="PGAT", in_dim=300, hidden_dim=600, out_dim=300, pos_dim=50, heads=[4, 1])),
    ("PGAT heads [8,1] hidden 256", dict(prop="PGAT", in_dim=250, hidden_dim=256, out_dim=500, pos_dim=50, heads=[8, 1])),
    ("PGAT hidden 128", dict(prop="PGAT", in_dim=250, hidden_dim=128, out_dim=128, pos_dim=50, heads=[4, 1])),
    ("PGAT hidden 250 pos 16 (4 column tiles)", dict(prop="PGAT", in_dim=250, hidden_dim=250, out_dim=500, pos_dim=16, heads=[4, 1])),
    ("PGAT in 768", dict(prop="PGAT", in_dim=768, hidden_dim=500, out_dim=500, pos_dim=50, heads=[4, 1])),
    ("PGAT heads [2,1] hidden 1000", dict(prop="PGAT", in_dim=250, hidden_dim=1000, out_dim=500, pos_dim=50, heads=[2, 1])),
    ("PGAT heads [1,1] hidden 500", dict(prop="PGAT", in_dim=250, hidden_dim=500, out_dim=250, pos_dim=50, heads=[1, 1])),
    ("PGCN hidden 1000", dict(prop="PGCN", in_dim=250, hidden_dim=1000, out_dim=500, pos_dim=50, heads=None)),
]
for name, c in CASES:
    tax = syn.make_taxonomy(29654, 46248, c["in_dim"], seed=47)
    torch.manual_seed(47)
    model = TaxoExpan(c["prop"], "WMR" if c["prop"] != "PGCN" else "MR", "LBM" if c["prop"] != "PGCN" else "BIM", in_dim=c["in_dim"], hidden_dim=c["hidden_dim"],
                      out_dim=c["out_dim"], pos_dim=c["pos_dim"], num_layers=1, heads=c["heads"], feat_drop=0.1, attn_drop=0.1, hidden_drop=0.1, out_drop=0.1).to(dev).train()
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
    batches = bench.build_batches(tax, 2, seed0=1000, device=dev)
    target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
    it = iter(range(10 ** 9))
    def one():
        bench.train_step(model, opt, batches[next(it) % 2], target, 1)
    dt = bench.median_time(one, reps=3, inner=10, warm=5)
    recs = [bench.profile_step(model, opt, b, target) for b in batches]
    roof = bench.summarize_profile(recs, [b["n_edges"] for b in batches], [b["n_nodes"] for b in batches], workload="pgat" if c["prop"] != "PGCN" else "pgcn")
    print(f"== {name}: {1e3 * dt:.3f} ms/step")
    for r in roof[:TOP]:
        print(f"     {r['kernel'][:58]:58s} {r['bound']:5s} frac {r['frac']:.2f}  {r['avg_us']:7.1f} us x {r['launches']}")
