#!/usr/bin/env python3
"""MAG-Full all-candidate encode (356 k egonets, 1.1 M nodes): median time and the library profiler's per-kernel view, with the
projected table rows formed inside the message/reduce sweep and (A/B switch) materialised first; checks that both agree bit for bit."""
import ctypes, os, sys, time, torch, numpy as np
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from taxoexpan_amd import synthetic as syn, graph as G, ops, _lib
from taxoexpan_amd.scoring import encode_candidates
dev = torch.device("cuda:0")
torch.manual_seed(47)
tax = syn.make_named_taxonomy("mag_full", seed=47)
model = bench.make_model("pgat", dev).eval()
cand, _v, _t = syn.split_candidates(tax)
dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
g = G.device_egonet_batch(dtax, cand, seed=7, with_features="lazy")
lib = _lib.load()
res = {}
with torch.no_grad():
    for sw in (False, True):
        ops._NO_TABLE_SWEEP = sw
        for _ in range(3): hg = encode_candidates(model, g)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t = time.perf_counter(); hg = encode_candidates(model, g); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        res[sw] = hg.clone()
        print("NO_TABLE_SWEEP", sw, "encode ms %.3f" % (sorted(ts)[3] * 1e3))
        lib.txe_profile_reset(); lib.txe_profile_enable(1)
        hg = encode_candidates(model, g); torch.cuda.synchronize()
        lib.txe_profile_enable(0)
        buf = ctypes.create_string_buffer(64); ms, work, kind = ctypes.c_float(), ctypes.c_double(), ctypes.c_int()
        for i in range(lib.txe_profile_count()):
            lib.txe_profile_get(i, buf, 64, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(kind))
            print("   %9.1f us  %7.2f %s  %s" % (ms.value * 1e3, work.value / (ms.value * 1e-3) / 1e12, "TB/s" if kind.value else "TF/s", buf.value.decode()))
        lib.txe_profile_reset()
print("bit-equal", torch.equal(res[False], res[True]), float((res[False] - res[True]).abs().max()))
