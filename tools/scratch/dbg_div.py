import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from taxoexpan_amd import ops, _lib, synthetic as syn
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = bench.make_model("pgat", dev)
from taxoexpan_amd.optim import Adam
opt = Adam(model.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
batches = bench.build_batches(tax, 4, seed0=1000, device=dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
print(bench.route_sanity(model, batches[0], target))
for i in range(330):
    loss = bench.train_step(model, opt, batches[i % 4], target, 1)
    if i % 10 == 9 or i > 280:
        l = float(loss.detach())
        fin = all(bool(torch.isfinite(p).all()) for p in model.parameters())
        print(i, l, "params finite", fin)
        if not fin: break
