import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from taxoexpan_amd import TaxoExpan, synthetic as syn
dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = TaxoExpan("PGAT", "WMR", "LBM", **bench.MAG).to(dev).train()
from taxoexpan_amd.optim import Adam  # noqa: E402
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
batches = bench.build_batches(tax, 2, 1000, dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
for i in range(3): bench.train_step(model, opt, batches[i % 2], target, 1)
torch.cuda.synchronize()
recs = bench.profile_step(model, opt, batches[0], target)
print("N", batches[0]["n_nodes"], "E", batches[0]["n_edges"])
for name, sec, work, kind in recs:
    print(f"{name:20s} {sec*1e6:9.1f} us  {'%.1f TF' % (work/sec/1e12) if kind==0 else '%.0f GB/s' % (work/sec/1e9)}  work={work:.3e}")
