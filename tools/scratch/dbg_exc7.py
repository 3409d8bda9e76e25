import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from taxoexpan_amd import ops, _lib, synthetic as syn
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "csrc", "libtxe.so"))
tax = syn.make_named_taxonomy("mag_cs", seed=47)
torch.manual_seed(47)
model = bench.make_model("pgat", dev)
batches = bench.build_batches(tax, 4, seed0=1000, device=dev)
target = torch.zeros(bench.N_QUERIES, dtype=torch.long, device=dev)
from taxoexpan_amd.optim import Adam
opt = Adam(model.parameters(), lr=1e-3, weight_decay=0, amsgrad=True)
out = (ctypes.c_int * 16)(); outf = (ctypes.c_float * 16)()
for i in range(400):
    loss = bench.train_step(model, opt, batches[i % 4], target, 1)
    if i % 50 != 49: continue
    torch.cuda.synchronize()
    lib.txe_debug_get(out, outf)
    if out[0]:
        print("step", i, "slow tiles", out[0], "first: tm,h,z,w,l,kbeg,kend,j,e =", list(out)[1:16], "val", outf[0], outf[1], "loss", float(loss))
        break
print("done", i)
