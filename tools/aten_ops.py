"""which torch (aten) GPU kernels still run inside one training step of the bench workload, and from where"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from taxoexpan_amd import TaxoExpan, synthetic as syn
from taxoexpan_amd.optim import Adam
dev = torch.device("cuda:0")
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = TaxoExpan("PGAT", "WMR", "LBM", **bench.MAG).to(dev).train()
opt = Adam(model.parameters(), lr=1e-3, amsgrad=True)
batches = bench.build_batches(tax, 2, 1000, dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
for i in range(3): bench.train_step(model, opt, batches[i % 2], target, 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.train_step(model, opt, batches[0], target, 1)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_device_time_total", row_limit=40, max_name_column_width=60, max_src_column_width=90))
