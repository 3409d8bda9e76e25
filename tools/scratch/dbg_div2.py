import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from taxoexpan_amd import ops, _lib, synthetic as syn
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
tax = syn.make_named_taxonomy("mag_cs", seed=47)
from taxoexpan_amd.optim import Adam
batches = bench.build_batches(tax, 4, seed0=1000, device=dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
for lr in (1e-3, 1e-4, 1e-5):
    torch.manual_seed(47)
    model = bench.make_model("pgat", dev)
    opt = Adam(model.parameters(), lr=lr, weight_decay=0, amsgrad=True)
    bench.route_sanity(model, batches[0], target)
    mx = 0.0
    for i in range(4000):
        loss = bench.train_step(model, opt, batches[i % 4], target, 1)
        if i % 100 == 99:
            l = float(loss.detach()); mx = max(mx, l)
            fin = all(bool(torch.isfinite(p).all()) for p in model.parameters())
            if not fin:
                print("lr", lr, "NaN params at step", i); break
    else:
        print("lr", lr, "finite after 4000 steps; last loss", l, "max sampled loss", mx)
