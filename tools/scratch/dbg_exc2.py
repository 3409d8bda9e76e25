import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from taxoexpan_amd import ops, _lib, synthetic as syn, graph as G
from taxoexpan_amd.scoring import encode_candidates
dev = torch.device("cuda:0")
def exc(x):
    t = x.abs().double()
    tiny = int(((t != 0) & (t < 2.0**-100)).sum()); huge = int((t >= 2.0**120).sum()); nf = int((~torch.isfinite(x)).sum())
    sub = int(((t != 0) & (t < 1.1754944e-38)).sum())
    mn = float(t[t != 0].min()) if (t != 0).any() else 0.0
    return dict(tiny=tiny, subnormal=sub, huge=huge, nonfinite=nf, min_nonzero=mn, numel=x.numel())
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = bench.make_model("pgat", dev)
batches = bench.build_batches(tax, 1, seed0=1000, device=dev)
b = batches[0]
print("x", exc(b["x"])); print("qf", exc(b["qf"]))
for k, p in model.named_parameters(): print(k, exc(p.data))
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
from taxoexpan_amd.optim import Adam
opt = Adam(model.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
with ops.debug_capture() as runs:
    bench.train_step(model, opt, b, target, 1)
torch.cuda.synchronize()
for k, p in model.named_parameters(): print("grad", k, exc(p.grad))
model.eval()
with torch.no_grad():
    cand, _v, test = syn.split_candidates(tax)
    dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
    g = G.device_egonet_batch(dtax, cand, seed=7, with_features="lazy")
    hg = encode_candidates(model, g)
    print("hg", exc(hg))
    U = ops.bilinear_project(hg, model.match.W.weight)
    print("U", exc(U))
    q = tax.features[torch.from_numpy(test[:1024])].to(dev)
    print("Q", exc(q))
