import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from taxoexpan_amd import ops, _lib, synthetic as syn
dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = bench.make_model("pgat", dev)
batches = bench.build_batches(tax, 2, seed0=1000, device=dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
from taxoexpan_amd.optim import Adam
opt = Adam(model.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
for i in range(30):
    loss = bench.train_step(model, opt, batches[i % 2], target, 1)
    if i % 5 == 0: print(i, float(loss))
print("params finite:", all(bool(torch.isfinite(p).all()) for p in model.parameters()))
recs = bench.profile_step(model, opt, batches[0], target)
for r in recs:
    print(r)
