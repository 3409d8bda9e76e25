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
from taxoexpan_amd.loss import info_nce_loss
opt = Adam(model.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
def exc(x):
    t = x.abs().double()
    return dict(huge=int((t >= 2.0**120).sum()), nonfinite=int((~torch.isfinite(x)).sum()), shape=tuple(x.shape), stride=x.stride())
b = batches[0]
g = b["g"]; g.ndata["pos"] = b["pos"]
with ops.debug_capture() as runs:
    pred = model(g, b["x"], b["qf"])
    _csr, _cfg, states = runs[0]
    for l, st in enumerate(states):
        for name in ("X", "Y", "Wp"):
            t = getattr(st, name, None)
            if torch.is_tensor(t):
                print(l, name, exc(t))
                if t.dim() == 2 and t.shape[1] > 300 and name == "X":
                    print("   X[:,300:] ", exc(t[:, 300:]), t[:3, 296:].tolist())
        print(l, [k for k in vars(st).keys()] if hasattr(st, "__dict__") else dir(st)[:40])
    loss = info_nce_loss(pred.reshape(bench.N_QUERIES, -1), target)
    loss.backward()
torch.cuda.synchronize()
recs = bench.profile_step(model, opt, batches[0], target)
print([r for r in recs if "split" in r[0]])
