"""training steps of an arbitrary (propagation, readout, matcher) combination on the bench batches -- run under rocprofv3 to look for
kernels that are out of proportion in the less-travelled variants:  python tools/variant_steps.py PGAT CR MLP"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taxoexpan_amd import TaxoExpan, synthetic as syn  # noqa: E402
from taxoexpan_amd.loss import info_nce_loss  # noqa: E402
from taxoexpan_amd.optim import Adam  # noqa: E402

prop, readout, match = sys.argv[1:4]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = TaxoExpan(prop, readout, match, **bench.MAG).to(dev).train()
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
batches = bench.build_batches(tax, 2, 1000, dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
import time
for i in range(steps + 3):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    b = batches[i % 2]
    b["g"].ndata["pos"] = b["pos"]
    opt.zero_grad(set_to_none=True)
    pred = model(b["g"], b["x"], b["qf"])
    loss = info_nce_loss(pred.reshape(bench.N_QUERIES, -1), target)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print(prop, readout, match, "ms/step", 1e3 * (time.perf_counter() - t0) / steps)
