#!/usr/bin/env python3
"""bench.py -- egonet-edges/s of the PGAT training step (BASELINE.json metric, configs[1]) on N MI355X of one node.

A "step" = one pass of the hot path over one batch: 128 queries x (1 positive + 31 negative) = 4,096 egonets of the
MAG-CS-shaped synthetic taxonomy per GPU (config_files/config.mag.json:28-30): PGAT (250+50 -> 4x500 -> 500) +
WeightedMeanReadout + LBM, InfoNCE, backward, Adam(amsgrad) -- exactly trainer/trainer.py:45-61 -- in training mode
(dropout 0.1 as the config says).  Inputs (CSR, features, queries) are resident in HBM before the timed region.
N > 1: one process per GPU (torch.distributed / RCCL), each rank its own batch (weak scaling), one flat gradient
all-reduce per step.

Prints ONE JSON line (rank 0) with value = total egonet-edges/s over all ranks, plus
  roofline      -- the dominant kernel by time in an instrumented pass (HIP events on the launch stream, per launch); flops are
                   ALGORITHMIC (unpadded operand sizes)
  roofline_all  -- the same for every kernel class (the message/reduce kernels are HBM-bound, the projections MFMA-bound)
  cpu_baseline  -- the CPU oracle (torch fp32, explicit COO; a restatement of the reference's DGL-CPU path, which cannot run: DGL 0.4
                   is not installable) timed on the host cores with BASELINE.md 2.3's protocol (3 warm-up + 10 timed, median) on
                   bounded samples: `step` (= the top-level value: fwd + InfoNCE + bwd), `fwd` (PGAT forward on a MAG-Full-shaped
                   egonet batch, the north star's >=10x target) and `scoring` (the literal per-query bilinear loop of
                   test_fast.py:121-123, and its factored form)
  extra         -- every secondary number as a median of >= 5 repetitions: forward-only edges/s (training batches and MAG-Full-shaped
                   batches, with the GPU/CPU ratio), the eval scoring loop on the MAG-CS and MAG-Full shapes (candidates scored / s),
                   and the training step of BASELINE configs[4] (PGCN+MR+BIM) and configs[3] (PGAT 2-layer on the MAG-Full taxonomy)
                   with their dominant kernel's roofline
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MAG = dict(in_dim=250, hidden_dim=500, out_dim=500, pos_dim=50, num_layers=1, heads=[4, 1], feat_drop=0.1, attn_drop=0.1,
           hidden_drop=0.1, out_drop=0.1)
N_QUERIES, NEG = 128, 31
# config.mag.json:66-73 trains with lr 1e-3 -- on real embeddings.  On the synthetic random-embedding taxonomy that rate drives the LBM's
# exp scores to overflow: loss 1e13 by step ~200 and NaN parameters by step ~240-300 (tools/ measurement, rounds 1-5 timed such steps
# unknowingly: the step's work does not depend on the values).  1e-4 trains the same model stably (loss 455 -> 79 over 4,000 steps); every
# timed model is checked finite at the end (assert_finite) so that a diverged run can never be reported as a number.
LR = 1e-4
PEAK_MFMA_F32 = 157.3e12      # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak, no TF32 on gfx950
PEAK_MFMA_BF16 = 2.5e15       # ... dense bf16 MFMA
SPLIT_PRODUCTS = 6            # csrc/txe_gemm_split.h: an fp32 product = six bf16 plane products (fp32-accurate) -> roof 2.5 PF / 6 per fp32 flop


def score_peak():
    """the roof of the scoring product: on the bf16 pipe (six plane products per fp32 product) unless the route is switched off"""
    from taxoexpan_amd import ops
    return PEAK_MFMA_F32 if ops._NO_SPLIT_GEMM else PEAK_MFMA_BF16 / SPLIT_PRODUCTS


def mfma_peak(kernel_name):
    """the matrix-pipe roof a kernel's ALGORITHMIC fp32 flops are priced against: the fp32 MFMA's, or -- for the products that run as six
    bf16 plane products per fp32 product (gemm_*_split_kernel) -- a sixth of the bf16 pipe's"""
    return PEAK_MFMA_BF16 / SPLIT_PRODUCTS if "_split_kernel" in kernel_name else PEAK_MFMA_F32
PEAK_HBM = 8.0e12             # spec; ~6.3e12 achievable


def hbm_copy_ceiling(device, mb=1024):
    """bytes/s (read + written) of a device copy of a buffer four times the Infinity Cache: what this box's HBM gives a streaming kernel
    with a 1:1 read/write mix -- the practical ceiling beside the 8 TB/s spec the HBM-bound kernels are priced against.  Returns
    (libtxe's 16-byte copy kernel txe_copy_stream -- the guide's ~6.3 TB/s figure --, torch's Tensor.copy_ for comparison)."""
    from taxoexpan_amd import _lib
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device=device).normal_()
    y = torch.empty_like(x)
    t_lib = median_time(lambda: _lib.call("txe_copy_stream", x.data_ptr(), y.data_ptr(), 4 * n, _lib.stream_ptr()), reps=5, inner=10, warm=2)
    t_torch = median_time(lambda: y.copy_(x), reps=5, inner=10, warm=2)
    return 2.0 * 4.0 * n / t_lib, 2.0 * 4.0 * n / t_torch


def median_time(fn, reps=5, inner=1, warm=1):
    """median wall time of `inner` calls of fn over `reps` repetitions (device synchronised around each repetition)"""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / inner)
    return float(np.median(ts))


def cpu_median(fn, budget_s=8.0):
    """3 warm-up + 10 timed iterations, median -- cut to 1 + 5 (and said so) when that would not fit the budget"""
    t0 = time.perf_counter()
    fn()
    first = time.perf_counter() - t0
    warm, iters = (3, 10) if first * 13 <= budget_s else (1, 5)
    for _ in range(warm - 1):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), f"{warm} warm-up + {iters} timed iterations, median"


def build_batches(tax, n_batches, seed0, device):
    from taxoexpan_amd import synthetic as syn
    out = []
    for b in range(n_batches):
        g, qf, labels = syn.training_batch(tax, N_QUERIES, NEG, seed=seed0 + b)
        x = g.ndata.pop("x").to(device)
        pos = g.ndata["pos"].to(device)
        g.csr(device)                    # CSR views resident in HBM
        out.append(dict(g=g, x=x, pos=pos, qf=qf.to(device), n_nodes=g.number_of_nodes(), n_edges=g.number_of_edges()))
    return out


def fresh_batch_begin(tax, dtax, seed, device, stream=None, repeated_queries=False):
    """a NEW training batch, first half (data_loaders.begin_device_batch): anchors sampled on the host (the reference's sampler is host
    Python too), uploaded, the egonets' node counts launched on `stream` -- nothing waited for"""
    from taxoexpan_amd.data_loaders import begin_device_batch
    rs = _RNG.setdefault("rs", np.random.RandomState(4711))             # (one generator for the run: seeding one costs 0.1 ms)
    has_par = _HAS_PAR.setdefault(id(tax), np.nonzero(np.diff(tax.par_ptr) > 0)[0])
    queries = has_par[rs.randint(0, len(has_par), size=N_QUERIES)]
    first = tax.par_ptr[queries]
    span = tax.par_ptr[queries + 1] - first
    anchors = rs.randint(0, tax.n_nodes, size=(N_QUERIES, 1 + NEG))       # column 0: a true parent; the rest: negatives
    anchors[:, 0] = tax.par_idx[first + (rs.random_sample(N_QUERIES) * span).astype(np.int64)]
    anchors = anchors.reshape(-1)
    exclude = np.full((N_QUERIES, 1 + NEG), -1, dtype=np.int64)
    exclude[:, 0] = queries
    return begin_device_batch(dtax, anchors, exclude.reshape(-1), np.repeat(queries, 1 + NEG), expand_factor=50, seed=seed + 1, stream=stream,
                              repeated_queries=repeated_queries)


def fresh_batch(tax, dtax, seed, device, stream=None):
    """a NEW training batch built inside the step, as a train.py epoch does for every step (data_loaders.py:9-28 + dataset.py:404-437):
    egonets + both CSR views + the feature gathers on the device (data_loaders.build_device_batch; on `stream` the construction's one
    host sync does not wait for the running step)"""
    from taxoexpan_amd.data_loaders import finish_device_batch
    return finish_device_batch(fresh_batch_begin(tax, dtax, seed, device, stream), dtax.features)


_HAS_PAR = {}
_RNG = {}
SECOND_STREAM_TAG = " [second stream]"


def train_step(model, opt, batch, target, world, loss_fn=None):
    from taxoexpan_amd.loss import info_nce_loss
    from taxoexpan_amd.scoring import allreduce_gradients
    g = batch["g"]
    g.ndata["pos"] = batch["pos"]
    opt.zero_grad(set_to_none=True)
    pred = model(g, batch["x"], batch["qf"])                       # trainer.py:51
    loss = (loss_fn or info_nce_loss)(pred.reshape(N_QUERIES, -1), target)   # trainer.py:52-56, loss.py:52-57 (one launch, gradient included)
    if world > 1:      # the output layer's gradient bucket is all-reduced under the backward of the layer below, the rest afterwards
        from taxoexpan_amd.scoring import overlapped_gradient_allreduce
        with overlapped_gradient_allreduce(model=model) as ov:
            loss.backward()                                              # trainer.py:60
        allreduce_gradients(list(model.parameters()), skip=ov)
    else:
        loss.backward()                                                  # trainer.py:60
    opt.step()                                                       # trainer.py:61
    return loss


def assert_finite(model, loss, what):
    """the timed steps trained a model whose loss and parameters are still finite (a diverged run times something else -- and, with
    non-finite operands, the bf16-pipe products recompute their tiles in fp32: csrc/txe_gemm_split.h)"""
    ok = bool(torch.isfinite(loss.detach()).all()) and all(bool(torch.isfinite(p).all()) for p in model.parameters())
    assert ok, f"{what}: the model diverged (non-finite loss or parameters) -- the timing is void"
    return float(loss.detach())


SETTLE_STEPS = 256        # untimed steps before the contract's --warmup (reported as config.settle_steps)


def route_sanity(model, batch, target):
    """the loss of one forward pass on the default route (the graph vector folded into the matcher where that applies) and on the plain
    one (ops._NO_MATCH_FOLD: hg = Z W^T formed) with the SAME dropout seeds: finite, and equal to 1e-4 relative -- a number timed on a
    route that computes something else would be worthless.  Returns what went into the JSON line (`config.sanity`)."""
    from taxoexpan_amd import ops
    from taxoexpan_amd.loss import info_nce_loss
    out = {}
    for name, off in (("default", False), ("no_match_fold", True)):
        prev, ops._NO_MATCH_FOLD = ops._NO_MATCH_FOLD, off
        try:
            ops.ROUTES.clear()
            torch.manual_seed(4711)                     # (ops.new_seed draws the dropout seeds from this generator)
            batch["g"].ndata["pos"] = batch["pos"]
            pred = model(batch["g"], batch["x"], batch["qf"])
            loss = info_nce_loss(pred.reshape(N_QUERIES, -1), target)
            out[name] = dict(loss=float(loss.item()), routes={k: ops.ROUTES.get(k) for k in ("match", "stack", "fold")})
        finally:
            ops._NO_MATCH_FOLD = prev
    a, b = out["default"]["loss"], out["no_match_fold"]["loss"]
    assert np.isfinite(a) and np.isfinite(b), out
    assert abs(a - b) <= 1e-4 * abs(b), f"the default route's loss differs from the plain route's: {out}"
    out["rel_diff"] = abs(a - b) / abs(b)
    return out


class ReferenceCaller(torch.nn.Module):
    """The reference's own model/model.py:70-87 forward -- four statements that know nothing of this library's routes -- over the three
    sub-modules of a taxoexpan_amd model: what a user gets who only swaps the import in model/model.py.  Timed as
    `step_reference_model_py_ms`; it must match `ms_per_step`."""

    def __init__(self, model):
        super().__init__()
        self.graph_propagate, self.readout, self.match = model.graph_propagate, model.readout, model.match

    def forward(self, g, h, qf):
        pos = g.ndata['pos'].to(h.device)
        g.ndata['h'] = self.graph_propagate(g, h)
        hg = self.readout(g, pos)
        prediction = self.match(hg, qf)
        return prediction


def profile_step(model, opt, batch, target):
    """one instrumented step: per-kernel durations from HIP events on the launch stream (libtxe profiling facility)"""
    from taxoexpan_amd import _lib
    lib = _lib.load()
    lib.txe_profile_reset()
    lib.txe_profile_enable(1)
    train_step(model, opt, batch, target, 1)
    torch.cuda.synchronize()
    lib.txe_profile_enable(0)
    recs = []
    buf = ctypes.create_string_buffer(64)
    ms, work, kind, strm = ctypes.c_float(), ctypes.c_double(), ctypes.c_int(), ctypes.c_void_p()
    main = torch.cuda.current_stream().cuda_stream
    for i in range(lib.txe_profile_count()):
        lib.txe_profile_get(i, buf, 64, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(kind))
        lib.txe_profile_stream(i, ctypes.byref(strm))
        # launches on the second stream run UNDER main-stream kernels: their own duration is stretched by sharing the machine
        name = buf.value.decode() + ("" if (strm.value or 0) == main else SECOND_STREAM_TAG)
        recs.append((name, ms.value * 1e-3, work.value, kind.value))
    lib.txe_profile_reset()
    return recs


def load_traffic(workload="pgat"):
    """HBM-side bytes per launch of each kernel from the committed rocprofv3 PMC passes (profiles/*_traffic.json, made by
    tools/summarize_profiles.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of the same workload, FETCH_SIZE
    corrected x2 as calibrated on a 1 GiB copy -- MI355X_MICROARCH.md HBM section).  PMC collection cannot run inside the
    timed process, so bench.py reports the latest committed measurement."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"*_{workload}_traffic.json")))
    if not files and workload == "pgat":
        files = sorted(f for f in glob.glob(os.path.join(REPO, "profiles", "*_traffic.json"))
                       if not any(t in os.path.basename(f) for t in ("_pgcn_", "_pgat2_", "_infer_")))
    if not files:
        return {}, None
    d = json.load(open(files[-1]))
    out = {}
    for kind in ("fetch", "write"):
        for k, v in d.get(kind, {}).get("kernels", {}).items():
            out.setdefault(k, 0.0)
            out[k] += v["avg_corrected_bytes"]
    return out, os.path.basename(files[-1])


def summarize_profile(all_recs, n_edges_by_launch, n_nodes_by_launch=None, workload="pgat"):
    """aggregate records by kernel name: launches, avg duration, algorithmic work per launch, roofline fraction"""
    traffic, traffic_src = load_traffic(workload)
    agg = {}
    n_graphs = N_QUERIES * (1 + NEG)
    for idx, (recs, e) in enumerate(zip(all_recs, n_edges_by_launch)):
        n = n_nodes_by_launch[idx] if n_nodes_by_launch else 0
        for name, sec, work, kind in recs:
            if name.startswith("readout_"):                 # the launcher knows G*D only: add the N*D rows read (+ written in bwd)
                work += (work / n_graphs) * n * (2 if "bwd" in name else 1)
            if kind == 1 and name.startswith("gat_"):       # add the E-proportional compulsory bytes (alpha/dz + CSR col)
                H = 4 if work > 4.0 * 2 * 1000 * 1000 else 1   # layer-0 (H=4, F=2000) vs layer-1 (H=1, F=500) rows
                work += 4.0 * e * (H + 1)
            a = agg.setdefault(name, dict(launches=0, sec=0.0, work=0.0, kind=kind))
            a["launches"] += 1
            a["sec"] += sec
            a["work"] += work
    out = []
    for name, a in agg.items():
        peak = PEAK_HBM if a["kind"] == 1 else mfma_peak(name)
        ach = a["work"] / a["sec"] if a["sec"] > 0 else 0.0
        second = name.endswith(SECOND_STREAM_TAG)
        name = name[:-len(SECOND_STREAM_TAG)] if second else name
        out.append(dict(kernel=name, stream="second (overlapped with main-stream kernels)" if second else "main",
                        bound="hbm" if a["kind"] == 1 else "mfma", launches=a["launches"],
                        avg_us=1e6 * a["sec"] / a["launches"], total_us=1e6 * a["sec"],
                        achieved=(ach / 1e9 if a["kind"] == 1 else ach / 1e12), peak=(peak / 1e9 if a["kind"] == 1 else peak / 1e12),
                        unit="GB/s" if a["kind"] == 1 else "TFLOP/s", frac=ach / peak, work_per_launch=a["work"] / a["launches"],
                        traffic=traffic.get(name), traffic_source=traffic_src if name in traffic else None))
        if a["kind"] != 1 and "_split_kernel" in name:
            out[-1]["pipe"] = "bf16 MFMA, %d plane products per fp32 product (fp32-accurate: DESIGN 4.10); peak = 2.5 PF/s / %d" % (SPLIT_PRODUCTS, SPLIT_PRODUCTS)
            out[-1]["frac_of_f32_mfma_peak"] = ach / PEAK_MFMA_F32
    out.sort(key=lambda r: -r["total_us"])
    return out


def concurrent_pair(all_recs, dom_kernel):
    """The dominant main-stream GEMM shares the chip with a second-stream GEMM launched right in front of it (the first layer's
    weight gradient and the skinny d_X product): each one's own duration is stretched by the other, so the pair is
    priced together -- flops of both over the longer of the two durations.  None if the dominant kernel runs alone."""
    flops = span = 0.0
    beside = None
    n = 0
    for recs in all_recs:
        for i, (name, sec, work, kind) in enumerate(recs):
            if name != dom_kernel or kind != 0 or i == 0:
                continue
            j = i - 1                                  # (its fix-up launch may sit in between)
            while j >= 0 and recs[j][0].endswith(SECOND_STREAM_TAG) and recs[j][3] != 0:
                j -= 1
            if j < 0 or not recs[j][0].endswith(SECOND_STREAM_TAG) or recs[j][3] != 0:
                continue
            pname, psec, pwork, pkind = recs[j]
            beside = pname[:-len(SECOND_STREAM_TAG)]
            flops += work + pwork
            span += max(sec, psec)
            n += 1
    if not n or span <= 0.0:
        return None
    ach = flops / span
    return {"kernel": beside, "stream": "second", "pair_flops_per_launch": flops / n, "pair_us": 1e6 * span / n,
            "pair_achieved": ach / 1e12, "unit": "TFLOP/s", "pair_frac": ach / PEAK_MFMA_F32}


def _oracle_graph(g, pos, n_g):
    """the first n_g egonets of a batch as the oracle's COO dict"""
    n_nodes = int(np.sum(g.batch_num_nodes[:n_g]))
    n_edges = int(np.sum(g.batch_num_edges[:n_g]))
    goff = torch.from_numpy(np.concatenate([[0], np.cumsum(g.batch_num_nodes[:n_g])])).long()
    return dict(src=torch.from_numpy(g._src[:n_edges]), dst=torch.from_numpy(g._dst[:n_edges]), pos=pos.cpu().long()[:n_nodes], graph_off=goff,
                num_nodes=n_nodes), n_nodes, n_edges


def cpu_baseline(batch, state_dict, full_batch=None, hg=None, queries=None, n_sample_queries=32):
    """The CPU oracle (oracle/txe_oracle.py: torch fp32 on explicit COO, all host threads; kind "port") on bounded samples.
    step:    forward + InfoNCE + backward on the first n_sample_queries x 32 egonets of training batch 0 (dropout as explicit masks);
    fwd:     PGAT+WMR+LBM forward (eval) on the first n_sample_queries x 32 egonets of a MAG-Full-shaped batch (`full_batch`);
    scoring: per query, the literal `match(hg, q.expand(G, -1))` of test_fast.py:121-123 over all G candidates (hg: the candidate
             vectors, encoded beforehand -- input data here), and the factored form S = Q (hg W)^T on a 64-query block."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import txe_oracle as orc
    out = {}
    n_g = n_sample_queries * (1 + NEG)
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in state_dict.items()}
    # ---- step ----
    graph, n_nodes, n_edges = _oracle_graph(batch["g"], batch["pos"], n_g)
    x, q = batch["x"][:n_nodes].cpu(), batch["qf"][:n_g].cpu()
    rs = np.random.RandomState(0)
    masks = []
    for l, (kt, H) in enumerate(((300, 4), (2050, 1))):      # dropout as explicit masks (same work as nn.Dropout)
        masks.append(dict(feat_keep=torch.from_numpy((rs.uniform(size=(n_nodes, kt)) >= 0.1).astype(np.float32)), feat_scale=1 / 0.9,
                          attn_keep=torch.from_numpy((rs.uniform(size=(n_edges, H, 1)) >= 0.1).astype(np.float32)), attn_scale=1 / 0.9))

    def step():
        for p in P.values():
            p.grad = None
        s, _, _ = orc.taxoexpan_forward(P, graph, x, q, "PGAT", "WMR", "LBM", [4, 1], 1, masks)
        orc.info_nce_loss(s, n_sample_queries).backward()
    # thread count: the fastest of {8, 16, 32, all} on one iteration each (the oracle's index_add / scatter kernels do not scale to
    # 128 threads: round 3's all-threads run was slower than BASELINE.md's 8-core anchor) -- every leg below runs at that count
    all_threads = torch.get_num_threads()
    step()
    sweep = {}
    for nt in sorted({min(n, all_threads) for n in (8, 16, 32, all_threads)}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        step()
        sweep[nt] = time.perf_counter() - t0
    best_nt = min(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    out["thread_sweep_s_per_iter"] = {str(k): v for k, v in sweep.items()}
    dt, proto = cpu_median(step)
    out["step"] = dict(value=n_edges / dt, unit="egonet-edges/s", s_per_iter=dt,
                       sample=f"{n_g} of the 4096 egonets of training batch 0 ({n_nodes} nodes, {n_edges} edges), fwd+InfoNCE+bwd, {proto}")
    # ---- forward only, MAG-Full-shaped egonets ----
    if full_batch is not None:
        graph_f, nn_f, ne_f = _oracle_graph(full_batch["g"], full_batch["pos"], n_g)
        xf, qf_ = full_batch["x"][:nn_f].cpu(), full_batch["qf"][:n_g].cpu()
        Pd = {k: v.detach() for k, v in P.items()}

        def fwd():
            with torch.no_grad():
                orc.taxoexpan_forward(Pd, graph_f, xf, qf_, "PGAT", "WMR", "LBM", [4, 1], 1, None)
        dt, proto = cpu_median(fwd)
        out["fwd"] = dict(value=ne_f / dt, unit="egonet-edges/s", s_per_iter=dt,
                          sample=f"{n_g} egonets of a MAG-Full-shaped (431,416-node taxonomy) batch ({nn_f} nodes, {ne_f} edges), "
                                 f"PGAT+WMR+LBM eval forward, {proto}")
    # ---- scoring loop ----
    if hg is not None:
        hg_c, W = hg.detach().cpu(), P["match.W.weight"].detach()
        G = hg_c.shape[0]
        qs = queries.cpu()
        it = iter(range(10 ** 9))

        def literal():                                          # one query of test_fast.py:121-123
            qv = qs[next(it) % qs.shape[0]]
            with torch.no_grad():
                orc.bilinear_match(hg_c, qv.expand(G, -1), W, True)
        dt, proto = cpu_median(literal, budget_s=4.0)
        qb = qs[:64]

        def factored():
            with torch.no_grad():
                torch.exp(qb @ (hg_c @ W[0]).t())
        dt2, proto2 = cpu_median(factored, budget_s=4.0)
        out["scoring"] = dict(value=G / dt, unit="candidates scored/s", s_per_query=dt, factored_value=G * qb.shape[0] / dt2,
                              sample=f"{G} MAG-CS candidates: literal per-query loop (one query per iteration, {proto}); factored form on a "
                                     f"{qb.shape[0]}-query block incl. U = hg W ({proto2})")
    top = dict(out["step"])
    top.update(cores=torch.get_num_threads(), host_threads_available=all_threads, kind="port",
               implementation="oracle/txe_oracle.py (torch CPU fp32)", **{k: v for k, v in out.items()})
    torch.set_num_threads(all_threads)
    return top


def _positives(tax, cand, test):
    cand_index = np.full(tax.n_nodes, -1, dtype=np.int64)
    cand_index[cand] = np.arange(len(cand))
    pos_lists = [cand_index[tax.par_idx[tax.par_ptr[q]:tax.par_ptr[q + 1]]] for q in test]
    pos_lists = [p[p >= 0] for p in pos_lists]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pos_lists])])
    idx = np.concatenate(pos_lists) if len(pos_lists) else np.zeros(0, dtype=np.int64)
    return off, idx


def extra_metrics(model, tax, device, batches, full_batches):
    """forward-only throughput and the eval scoring loop (candidates scored / s) on one GPU; every time a median of 5 repetitions.
    Returns (metrics, hg of the MAG-CS candidates, their test queries) -- the latter two feed cpu_baseline's scoring leg."""
    from taxoexpan_amd import graph as G, ops, synthetic as syn
    from taxoexpan_amd.scoring import encode_candidates, rank_all_fused, score_all
    out = {"timing": "median of 5 repetitions each (device synchronised around every repetition)"}
    model.eval()
    with torch.no_grad():
        def fwd_all(bs):
            for b in bs:
                b["g"].ndata["pos"] = b["pos"]
                model(b["g"], b["x"], b["qf"])
        dt = median_time(lambda: fwd_all(batches), reps=5, inner=2, warm=2)
        out["pgat_fwd_eval_edges_per_s"] = sum(b["n_edges"] for b in batches) / dt
        if full_batches:
            dt = median_time(lambda: fwd_all(full_batches), reps=5, inner=2, warm=2)
            out["pgat_fwd_eval_mag_full_batches_edges_per_s"] = sum(b["n_edges"] for b in full_batches) / dt
        # all-candidate inference (test_fast.py small mode): encode every candidate egonet, score every test query
        cand, _val, test = syn.split_candidates(tax)
        g = syn.egonet_batch(tax, cand, seed=7)
        g.ndata["x"] = g.ndata["x"].to(device)
        g.csr(device)
        queries = tax.features[torch.from_numpy(test)].to(device)
        t_enc = median_time(lambda: encode_candidates(model, g), warm=2)
        hg = encode_candidates(model, g)
        # the same candidates as device-built egonets whose features stay rows of the taxonomy table (evaluate.py's path): the
        # layer-0 projection runs once per taxonomy node
        dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, device)
        gl = G.device_egonet_batch(dtax, cand, seed=7, with_features="lazy")
        t_enc_l = median_time(lambda: encode_candidates(model, gl), warm=2)
        out["infer_encode_dedup_s"] = t_enc_l
        out["infer_encode_dedup_edges_per_s"] = gl.number_of_edges() / t_enc_l
        del gl
        S = score_all(model.match, hg, queries)
        t_sc = median_time(lambda: score_all(model.match, hg, queries, out=S))
        off_np, idx_np = _positives(tax, cand, test)
        off, idx = torch.tensor(off_np, dtype=torch.int32), torch.tensor(idx_np, dtype=torch.int32)
        t_rk = median_time(lambda: ops.rank_block(S, off, idx, True))
        ranks = ops.rank_block(S, off, idx, True)
        # fused scoring + ranking (no score matrix): must give the same ranks
        t_fused = median_time(lambda: rank_all_fused(model.match, hg, queries, off, idx))
        ranks_f = rank_all_fused(model.match, hg, queries, off, idx)
        pairs = float(len(cand)) * len(test)
        out["fused_rank_equals_materialised"] = bool(torch.equal(ranks_f.cpu(), ranks.cpu()))
        out.update(infer_candidates=int(len(cand)), infer_queries=int(len(test)), infer_encode_s=t_enc,
                   infer_encode_edges_per_s=g.number_of_edges() / t_enc, infer_score_s=t_sc, infer_rank_s=t_rk,
                   infer_fused_score_rank_s=t_fused, candidates_scored_per_s=pairs / t_sc,
                   candidates_scored_per_s_incl_encode_and_rank=pairs / (t_enc + t_sc + t_rk),
                   candidates_scored_per_s_fused_rank_incl_dedup_encode=pairs / (t_enc_l + t_fused),
                   score_gemm_tflops=2.0 * 250 * pairs / t_sc / 1e12, score_gemm_frac_of_mfma_peak=2.0 * 250 * pairs / t_sc / score_peak(), score_gemm_frac_of_f32_mfma_peak=2.0 * 250 * pairs / t_sc / PEAK_MFMA_F32,
                   # (SURVEY 8d's factored count of the SAME timed region: U = hg W once, 2 G l r, + 2 r per pair)
                   score_factored_tflops=(2.0 * 250 * pairs + 2.0 * hg.shape[0] * 500 * 250) / t_sc / 1e12,
                   score_factored_frac_of_mfma_peak=(2.0 * 250 * pairs + 2.0 * hg.shape[0] * 500 * 250) / t_sc / score_peak(),
                   mean_rank=float(ranks.float().mean().item()) if ranks.numel() else None)
    model.train()
    return out, hg, queries[:256]


def extra_metrics_mag_full(model, device, tax, n_queries=8192, qblock=1024):
    """N = 1: the all-candidate inference loop on the MAG-Full shape (356 k candidate egonets built on device, features as rows of
    the taxonomy table; 8,192 of the test queries): encode in one batch and in test_fast.py's `-b 30000` chunks, then score + rank
    every (query, candidate) pair -- the `candidates scored / s` half of BASELINE.json's metric at the size it is quoted on."""
    from taxoexpan_amd import graph as G, ops as ops_, synthetic as syn
    from taxoexpan_amd.evaluate import candidate_graphs
    from taxoexpan_amd.scoring import encode_candidates, rank_all_fused
    out = {}
    model.eval()
    with torch.no_grad():
        cand, _val, test = syn.split_candidates(tax)
        test = test[:n_queries]
        dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, device)
        t_build = median_time(lambda: G.device_egonet_batch(dtax, cand, seed=7, with_features="lazy"))
        g = G.device_egonet_batch(dtax, cand, seed=7, with_features="lazy")
        queries = tax.features[torch.from_numpy(test)].to(device)
        t_enc = median_time(lambda: encode_candidates(model, g), warm=2)          # (warm-up: the caching allocator grows by ~30 GB)
        hg = encode_candidates(model, g)
        n_nodes, n_edges = int(g.number_of_nodes()), int(g.number_of_edges())
        del g
        chunks = candidate_graphs(dtax, cand, 50, 7, batch_size=30000)            # BASELINE configs[2]: batch_size=30000
        t_enc_c = median_time(lambda: encode_candidates(model, chunks), warm=2)
        n_chunks = len(chunks)
        del chunks
        t_sc = median_time(lambda: _score_local_only(model, hg, queries, qblock))
        pos_off, pos_idx = _positives(tax, cand, test)
        t_fr = median_time(lambda: rank_all_fused(model.match, hg, queries, pos_off, pos_idx, block=qblock))
        ranks = rank_all_fused(model.match, hg, queries, pos_off, pos_idx, block=qblock)
        # infer.py:96-106 / test_fast.py:121-131: the 5 best parents of every query -- the fused score + select kernels against the torch
        # composite on materialised score blocks (scoring.topk_parents; on 1,024 of the queries: it needs two int64 [Q, G] temporaries)
        from taxoexpan_amd.scoring import topk_parents, topk_parents_fused
        t_top = median_time(lambda: topk_parents_fused(model.match, hg, queries, None, 5, True, block=qblock))
        U = ops_.bilinear_project(hg, model.match.W.weight)
        ids = torch.arange(hg.shape[0], device=device)
        qs = queries[:1024]

        def composite():
            return topk_parents(ops_.score_block(qs, U, model.match.apply_exp), ids, 5, True)
        t_comp = median_time(composite, reps=3)
        same = bool(torch.equal(topk_parents_fused(model.match, hg, qs, None, 5, True, block=qblock), composite()))
        del U
        out.update(infer_top5_s=t_top, infer_top5_queries_per_s=len(test) / t_top, infer_top5_pairs_per_s=float(len(cand)) * len(test) / t_top,
                   infer_top5_torch_composite_queries_per_s=qs.shape[0] / t_comp, infer_top5_fused_over_composite=(len(test) / t_top) / (qs.shape[0] / t_comp),
                   infer_top5_fused_equals_composite=same)
        pairs = float(len(cand)) * len(test)
        out.update(shape="mag_full", candidates=int(len(cand)), queries=int(len(test)), egonet_nodes=n_nodes, egonet_edges=n_edges,
                   device_egonet_build_s=t_build, encode_s=t_enc, encode_edges_per_s=n_edges / t_enc,
                   encode_30000_chunks_s=t_enc_c, encode_30000_chunks_edges_per_s=n_edges / t_enc_c, encode_chunks=n_chunks,
                   score_s=t_sc, candidates_scored_per_s=pairs / t_sc,
                   score_gemm_tflops=2.0 * 250 * pairs / t_sc / 1e12, score_gemm_frac_of_mfma_peak=2.0 * 250 * pairs / t_sc / score_peak(), score_gemm_frac_of_f32_mfma_peak=2.0 * 250 * pairs / t_sc / PEAK_MFMA_F32,
                   # (SURVEY 8d's factored count of the SAME timed region: U = hg W once, 2 G l r, + 2 r per pair)
                   score_factored_tflops=(2.0 * 250 * pairs + 2.0 * hg.shape[0] * 500 * 250) / t_sc / 1e12,
                   score_factored_frac_of_mfma_peak=(2.0 * 250 * pairs + 2.0 * hg.shape[0] * 500 * 250) / t_sc / score_peak(),
                   fused_score_rank_s=t_fr, candidates_scored_per_s_fused_rank=pairs / t_fr,
                   candidates_scored_per_s_fused_rank_incl_encode=pairs / (t_fr + t_enc),
                   mean_rank=float(ranks.float().mean().item()) if ranks.numel() else None)
    model.train()
    return out


def make_model(workload, device):
    from taxoexpan_amd import TaxoExpan
    if workload == "pgat":
        return TaxoExpan("PGAT", "WMR", "LBM", **MAG).to(device).train()
    if workload == "pgat2":
        return TaxoExpan("PGAT", "WMR", "LBM", **dict(MAG, num_layers=2, heads=[4, 4, 1])).to(device).train()
    if workload == "semeval":      # BASELINE configs[0]: config.wordnet.json's dimensions (in 300, hidden 600, out 300)
        return TaxoExpan("PGAT", "WMR", "LBM", **dict(MAG, in_dim=300, hidden_dim=600, out_dim=300)).to(device).train()
    return TaxoExpan("PGCN", "MR", "BIM", **MAG).to(device).train()


WORKLOAD_TEXT = {"pgat": "MAG-CS synthetic taxonomy (29,654 nodes, d=250), PGAT+WMR+LBM fp32 dims 250/50/500/500 heads [4,1], ",
                 "pgat2": "MAG-Full synthetic taxonomy (431,416 nodes, d=250), PGAT num_layers=2 +WMR+LBM fp32 dims 250/50/500/500 heads [4,4,1], ",
                 "pgcn": "MAG-CS synthetic taxonomy (29,654 nodes, d=250), PGCN+MR+BIM fp32 dims 250/50/500/500, ",
                 "semeval": "SemEval-Noun synthetic taxonomy (83,073 nodes, d=300), PGAT+WMR+LBM fp32 dims 300/50/600/300 heads [4,1], "}
STEP_TEXT = "128 queries x 32 = 4096 egonets per GPU per step, fwd + InfoNCE + bwd + Adam(amsgrad), dropout 0.1"


def variant_step(workload, tax, device, steps=10, reps=5):
    """the training step of another BASELINE config in the same process (configs[4] `pgcn`, configs[3]'s model `pgat2`): median ms per
    step over `reps` groups of `steps` steps, and the roofline of its dominant kernel from one instrumented step per batch"""
    from taxoexpan_amd.optim import Adam
    torch.manual_seed(47)
    model = make_model(workload, device)
    opt = Adam(model.parameters(), lr=LR, weight_decay=0, amsgrad=True)
    batches = build_batches(tax, 2, seed0=1000, device=device)
    target = torch.zeros(N_QUERIES, dtype=torch.long, device=device)
    it = iter(range(10 ** 9))

    def one():
        train_step(model, opt, batches[next(it) % 2], target, 1)
    dt = median_time(one, reps=reps, inner=steps, warm=5)
    assert_finite(model, train_step(model, opt, batches[0], target, 1), workload)
    edges = float(np.mean([b["n_edges"] for b in batches]))
    recs = [profile_step(model, opt, b, target) for b in batches]
    roof = summarize_profile(recs, [b["n_edges"] for b in batches], [b["n_nodes"] for b in batches], workload=workload)
    dom = next((r for r in roof if r["stream"] == "main"), roof[0])
    return dict(workload=WORKLOAD_TEXT[workload] + STEP_TEXT, ms_per_step=1e3 * dt, egonet_edges_per_s=edges / dt,
                timing=f"median of {reps} x {steps} steps",
                roofline={k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_us", "work_per_launch")},
                roofline_top5=[{k: r[k] for k in ("kernel", "bound", "frac", "avg_us", "launches", "total_us")} for r in roof[:5]])


def large_batch_step(tax, device, factor=8):
    """SURVEY 8d's steady-state point: the BASELINE configs[1] step on `factor` x 4,096 egonets per step (128 x factor queries x 32) -- the
    fixed per-launch costs of the ~30 launches are amortised, the kernels run near their large-batch rates"""
    global N_QUERIES
    base = N_QUERIES
    try:
        N_QUERIES = base * factor
        r = variant_step("pgat", tax, device, steps=5, reps=5)
        r["workload"] = WORKLOAD_TEXT["pgat"] + f"{N_QUERIES} queries x 32 = {N_QUERIES * 32} egonets per step, fwd + InfoNCE + bwd + Adam(amsgrad), dropout 0.1"
        r["ms_per_4096_egonets"] = r["ms_per_step"] / factor
        return r
    finally:
        N_QUERIES = base


def semeval_step(device):
    """BASELINE configs[0] on the GPU: the SemEval-Noun shape (config.wordnet.json's dimensions), 64 queries x 32 = 2,048 egonets per step
     -- the reference's CPU-runnable case; its parity at this size is tests/test_gpu_full_size.py [semeval-*]"""
    global N_QUERIES
    from taxoexpan_amd import synthetic as syn
    base = N_QUERIES
    try:
        N_QUERIES = 64
        tax = syn.make_named_taxonomy("semeval_noun", seed=47)
        r = variant_step("semeval", tax, device, steps=10, reps=5)
        r["workload"] = WORKLOAD_TEXT["semeval"] + "64 queries x 32 = 2048 egonets per step, fwd + InfoNCE + bwd + Adam(amsgrad), dropout 0.1"
        return r
    finally:
        N_QUERIES = base


def dp_step_pgat2(device, world, rank, tax_full, steps=10, reps=3):
    """BASELINE configs[3] as BASELINE names it: the 2-layer PGAT (heads [4,4,1]) on the MAG-Full-shaped taxonomy, data-parallel over
    queries -- 4,096 egonets per rank per step (config.mag.json:28-30), identical replicas, gradients all-reduced over RCCL with the
    output layer's bucket overlapped under the backward of the layers below (scoring.overlapped_gradient_allreduce), Adam on every rank.
    Barrier + synchronize on both sides of every timed group, MAX over ranks, median of `reps` groups; edges summed over the ranks."""
    from taxoexpan_amd.optim import Adam
    torch.manual_seed(47)
    model = make_model("pgat2", device)
    for p in model.parameters():
        dist.broadcast(p.data, src=0)
    opt = Adam(model.parameters(), lr=LR, weight_decay=0, amsgrad=True)
    batches = build_batches(tax_full, 2, seed0=7000 + 1000 * rank, device=device)
    target = torch.zeros(N_QUERIES, dtype=torch.long, device=device)
    for i in range(5):
        train_step(model, opt, batches[i % 2], target, world)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            train_step(model, opt, batches[i % 2], target, world)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    assert_finite(model, train_step(model, opt, batches[0], target, world), "pgat2 dp")
    t = torch.tensor(ts, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.median().item())
    e = torch.tensor([float(np.mean([b["n_edges"] for b in batches]))], dtype=torch.float64, device=device)
    dist.all_reduce(e, op=dist.ReduceOp.SUM)
    n_grad = sum(p.numel() for p in model.parameters())
    return dict(workload=WORKLOAD_TEXT["pgat2"] + STEP_TEXT + f", dp{world} gradient all-reduce", ms_per_step=1e3 * dt,
                egonet_edges_per_s=float(e.item()) / dt, timing=f"median of {reps} x {steps} steps, max over ranks",
                gradient_bytes_per_rank_per_step=4.0 * n_grad)


def extra_metrics_sharded(model, device, world, rank, n_queries=int(os.environ.get("TXE_BENCH_SHARDED_QUERIES", "8192")), qblock=1024):
    """N > 1: all-candidate inference on the MAG-Full shape, candidates sharded contiguously over the ranks (each rank encodes
    and scores its shard), score blocks all-gathered over xGMI so every rank holds the full [queries x candidates] block
    (north star).  Reports the compute-only and the all-gather-inclusive pair rates (max over ranks)."""
    from taxoexpan_amd import graph as G, synthetic as syn
    from taxoexpan_amd.scoring import encode_candidates, score_all_sharded, shard_bounds
    out = {}
    model.eval()
    with torch.no_grad():
        tax = syn.make_named_taxonomy("mag_full", seed=47)
        cand, _val, test = syn.split_candidates(tax)
        test = test[:n_queries]
        lo, hi = shard_bounds(len(cand), world, rank)
        dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, device)
        g = G.device_egonet_batch(dtax, cand[lo:hi], seed=7, with_features="lazy")    # (a shard smaller than the table gathers as usual)
        queries = tax.features[torch.from_numpy(test)].to(device)
        for _ in range(2):                                                 # warm-up (the caching allocator grows twice)
            hg = encode_candidates(model, g)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        hg = encode_candidates(model, g)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t_enc = time.perf_counter() - t0
        sink = lambda q0, blk: None            # (a consumer would index blk.shards [world, nq, c] in place)
        timings = {}
        for name, gather in (("local", False), ("allgather", True)):
            def run():
                if gather:
                    score_all_sharded(model.match, hg, len(cand), queries, block=qblock, on_block=sink)
                else:                                                       # same local work, no collective
                    _score_local_only(model, hg, queries, qblock)
            run()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            timings[name] = time.perf_counter() - t0
        # fused ranking: thresholds and counts all-reduced ([n_positives] vectors) instead of gathering the score blocks
        from taxoexpan_amd.scoring import rank_all_fused
        cand_index = np.full(tax.n_nodes, -1, dtype=np.int64)
        cand_index[cand] = np.arange(len(cand))
        pos_lists = [cand_index[tax.par_idx[tax.par_ptr[qn]:tax.par_ptr[qn + 1]]] for qn in test]
        pos_lists = [p[p >= 0] for p in pos_lists]
        pos_off = np.concatenate([[0], np.cumsum([len(p) for p in pos_lists])])
        pos_idx = np.concatenate(pos_lists) if pos_lists else np.zeros(0, dtype=np.int64)
        rank_all_fused(model.match, hg, queries, pos_off, pos_idx, block=qblock, shard_lo=lo, sharded=True)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ranks = rank_all_fused(model.match, hg, queries, pos_off, pos_idx, block=qblock, shard_lo=lo, sharded=True)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t_fr = time.perf_counter() - t0
        # the 5 best parents of every query (infer.py:96-106): each rank selects among its shard with the fused kernels, the [Q, 5] lists
        # (40 bytes per query and rank) are all-gathered and merged
        from taxoexpan_amd.scoring import topk_parents_fused
        topk_parents_fused(model.match, hg, queries, None, 5, True, block=qblock, shard_lo=lo, sharded=True)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        top5 = topk_parents_fused(model.match, hg, queries, None, 5, True, block=qblock, shard_lo=lo, sharded=True)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t_top = time.perf_counter() - t0
        t = torch.tensor([t_enc, timings["local"], timings["allgather"], t_fr, t_top], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_enc, t_loc, t_ag, t_fr, t_top = (float(x) for x in t.tolist())
        pairs = float(len(cand)) * len(test)
        out.update(infer_fused_rank_allreduce_s=t_fr, candidates_scored_per_s_fused_allreduce=pairs / t_fr,
                   infer_top5_allgather_s=t_top, infer_top5_queries_per_s_sharded=len(test) / t_top, infer_top5_shape=list(top5.shape),
                   mean_rank=float(ranks.float().mean().item()) if ranks.numel() else None)
        out.update(shape="mag_full", infer_candidates=int(len(cand)), infer_queries=int(len(test)), candidates_per_rank=int(hi - lo),
                   infer_encode_s=t_enc, infer_score_local_s=t_loc, infer_score_allgather_s=t_ag,
                   candidates_scored_per_s_local=pairs / t_loc, candidates_scored_per_s_allgather=pairs / t_ag,
                   candidates_scored_per_s_allgather_incl_encode=pairs / (t_ag + t_enc),
                   allgather_bytes_per_rank_per_block=4.0 * qblock * len(cand),
                   # each rank RECEIVES (world-1)/world of every [qblock, candidates] score block; the rate over the whole gathered loop
                   # (compute + collective, pipelined) -- a lower bound on what the links carried
                   allgather_gbs_per_rank=4.0 * pairs * (world - 1) / world / t_ag / 1e9,
                   allreduce_counts_queries_per_s=len(test) / t_fr)
    model.train()
    return out


def _score_local_only(model, hg, queries, qblock):
    from taxoexpan_amd import ops
    U = ops.bilinear_project(hg, model.match.W.weight)
    S = None
    for q0 in range(0, queries.shape[0], qblock):
        S = ops.score_block(queries[q0:q0 + qblock], U, model.match.apply_exp, out=None if S is None or S.shape[0] != min(qblock, queries.shape[0] - q0) else S)
    return S


COMPACT_LIMIT = 4096          # bytes: the driver keeps an 8 KB stdout tail and parses the LAST line of it (round 5's 21 KB line was cut)
COMPACT_SCALARS = ("step_fp32_mfma_ms", "step_repeated_queries_ms", "step_incl_batch_build_ms", "step_reference_model_py_ms",
                   "candidates_scored_per_s_mag_cs", "candidates_scored_per_s_mag_full", "candidates_scored_per_s_mag_full_fused_rank",
                   "mag_full_encode_edges_per_s", "pgat_fwd_eval_mag_full_edges_per_s", "gpu_over_cpu_pgat_fwd_mag_full",
                   "step_pgcn_ms", "step_pgat2_ms", "step_semeval_ms",
                   # N > 1 (flat copies of extra_metrics_sharded / dp_step_pgat2): what RCCL ran, on how many ranks
                   "rccl_world", "collective_backend", "step_pgat2_dp_ms", "step_pgat2_dp_edges_per_s",
                   "candidates_scored_per_s_allgather", "candidates_scored_per_s_fused_allreduce",
                   "allgather_gbs_per_rank", "allreduce_counts_queries_per_s")


def compact_line(full):
    """The ONE stdout line of the contract, <= COMPACT_LIMIT bytes: the driver's keys, `config` (workload, sizes, parallelism, routes),
    a flat `roofline`, a flat `cpu_baseline` and a handful of flat scalars.  Everything else bench.py measures (roofline_all, the A/B legs,
    the per-config extras, the prose) goes to bench_extra.json and to stderr -- never to stdout."""
    c, r, cb = full["config"], full["roofline"], full.get("cpu_baseline") or None
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    # `dtype` is the arithmetic the path computes in: fp32 in, fp32 out, fp32 accumulation.  `matrix_pipe` says HOW the big products are
    # formed and `value_fp32_mfma` is the same step with them on the fp32 MFMA instruction -- both numbers in one line
    line["matrix_pipe"] = full["matrix_pipe"]
    if full.get("step_fp32_mfma_ms"):
        line["value_fp32_mfma"] = full["value"] * full["ms_per_step"] / full["step_fp32_mfma_ms"]
    line["config"] = {k: c[k] for k in ("workload", "egonets_per_step_per_gpu", "avg_edges_per_step_per_gpu", "parallelism", "routes",
                                        "settle_steps", "lr", "final_loss") if k in c}
    line["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_us", "work_per_launch",
                                          "hbm_kernel", "hbm_frac", "hbm_avg_us", "hbm_achieved_gbs", "hbm_traffic_ratio",
                                          "hbm_frac_of_copy_ceiling", "copy_ceiling_gbs", "hbm_aggregate_fwd_frac", "hbm_fused_bwd_frac",
                                          "hbm_dx_pos_frac", "mfma_main_stream_frac") if k in r}
    if cb is not None:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "fwd_value", "fwd_unit", "scoring_value",
                                                   "scoring_unit", "scoring_factored_value") if k in cb}
        line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample", ""))[:240]
    else:
        line["cpu_baseline"] = None
    for k in COMPACT_SCALARS:
        if full.get(k) is not None:
            line[k] = full[k]
    line["extra_file"] = "bench_extra.json"

    def rnd(o):                      # 6 significant digits: the line stays short, nothing the driver checks is that fine
        if isinstance(o, float):
            return float(f"{o:.6g}")
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [rnd(v) for v in o]
        return o
    keep_exact = {k: line[k] for k in ("value", "ms_per_step")}
    line = rnd(line)
    line.update(keep_exact)
    line["config"]["avg_edges_per_step_per_gpu"] = c["avg_edges_per_step_per_gpu"]
    return line


def emit(full):
    """full record -> bench_extra.json (+ gpurun_out/ when it exists) and stderr; compact record -> the single stdout line"""
    text = json.dumps(full)
    for path in (os.path.join(REPO, "bench_extra.json"), os.path.join(REPO, "gpurun_out", "bench_extra.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(text + "\n")
        except OSError:
            pass
    print("BENCH_FULL " + text, file=sys.stderr, flush=True)
    out = json.dumps(compact_line(full), separators=(",", ":"))
    assert len(out) < COMPACT_LIMIT, f"compact bench line is {len(out)} bytes"
    print(out, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--workload", default="pgat", choices=["pgat", "pgcn", "pgat2"],
                    help="pgat = BASELINE configs[1] (default, the metric's workload); pgcn = configs[4] (PGCN+MR+BIM, same batches); "
                         "pgat2 = configs[3] (MAG-Full-shaped taxonomy, PGAT num_layers=2 heads [4,4,1])")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if os.environ.get("TXE_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TXE_BENCH_BACKEND", "nccl")     # "nccl" = RCCL over xGMI; "gloo" only to exercise the N>1
        if backend == "nccl":                                       # logic on a single-GPU box (both ranks on cuda:0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # The backward of a step is a chain of three Functions on ONE device: autograd's per-device engine thread only adds a hand-off
    # (0.2 ms of HOST time per step, tools/fresh_batch_timing.py) -- the backward runs on the calling thread.  Same launches, same
    # numbers; TXE_ENGINE_THREADS=1 is the A/B switch.
    torch.autograd.set_multithreading_enabled(os.environ.get("TXE_ENGINE_THREADS", "0") == "1")

    from taxoexpan_amd import synthetic as syn
    tax = syn.make_named_taxonomy("mag_full" if args.workload == "pgat2" else "mag_cs", seed=47)
    torch.manual_seed(47)
    model = make_model(args.workload, device)
    if world > 1:                                   # identical replicas
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    from taxoexpan_amd.optim import Adam          # torch.optim.Adam's update (config.mag.json:66-73) as one HIP launch
    opt = Adam(model.parameters(), lr=LR, weight_decay=0, amsgrad=True)
    batches = build_batches(tax, 4, seed0=1000 * (rank + 1), device=device)
    target = torch.zeros(N_QUERIES, dtype=torch.long, device=device)
    sanity = route_sanity(model, batches[0], target)       # before anything is timed: the step's loss is finite and route-independent

    # before the contract's W warm-up steps: a quarter of a second of the same steps, untimed -- a process that starts on an idle GPU has
    # been seen to run its first ~150 steps 20 % slow (clocks and the caching allocator settling: 1.23 ms where every later leg of the
    # same run read 1.01 ms); the same count on every rank
    for i in range(SETTLE_STEPS):
        train_step(model, opt, batches[i % len(batches)], target, world)
    # the host enqueues a step in ~0.72 ms against ~0.95 ms on the GPU: a full (generation-2) pass of Python's cyclic collector over the
    # taxonomy's and the batches' objects -- ~100 ms, every few hundred steps (tools/host_stalls.py found one at step 346 of 400; none
    # with the collector off) -- inside the K timed steps would read as +2 ms per step.  Collect now, then keep the survivors out of
    # later passes (gc.freeze): the collector stays on, its passes stay short
    import gc
    gc.collect()
    gc.freeze()
    for i in range(args.warmup):
        train_step(model, opt, batches[i % len(batches)], target, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = 0
    for i in range(args.steps):
        b = batches[i % len(batches)]
        last_loss = train_step(model, opt, b, target, world)
        edges += b["n_edges"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    final_loss = assert_finite(model, last_loss, "timed steps")
    from taxoexpan_amd import ops as _ops_r
    routes_timed = {k: _ops_r.ROUTES.get(k) for k in ("match", "stack", "fold", "stack_bwd")}     # the routes the timed steps took
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        e = torch.tensor([edges], dtype=torch.float64, device=device)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        edges = float(e.item())

    # the same resident batches with the query features as ops.RepeatedRows -- one row per query + the runs of its 1 + NEG pairs, what
    # data_loaders.DeviceBatchLoader hands out -- instead of the reference collate's stack of one row per pair (data_loaders.py:9-28):
    # BIM / LBM then project 128 rows instead of 4,096.  Reported beside `value`, which stays on the reference's input format.
    rq_ms = None
    if args.workload == "pgat" and world == 1:
        from taxoexpan_amd import ops as _ops
        rq_batches = []
        for b in batches:
            q = b["qf"].cpu().numpy()
            rid = np.concatenate([[0], np.cumsum(np.any(q[1:] != q[:-1], axis=1))])
            first = np.concatenate([[True], rid[1:] != rid[:-1]])
            rq_batches.append(dict(b, qf=_ops.RepeatedRows.from_ids(torch.from_numpy(q[first]).to(device), rid)))
        for i in range(min(args.warmup, 5)):
            train_step(model, opt, rq_batches[i % len(batches)], target, world)
        torch.cuda.synchronize()
        tq0 = time.perf_counter()
        for i in range(args.steps):
            train_step(model, opt, rq_batches[i % len(batches)], target, world)
        torch.cuda.synchronize()
        rq_ms = 1e3 * (time.perf_counter() - tq0) / max(args.steps, 1)

    # A/B scalars of the three caller-side choices `value` rests on beside the reference's own train.py: autograd's
    # per-device engine thread (bench.py runs backward on the calling thread), torch.optim.Adam instead of taxoexpan_amd.optim.Adam,
    # torch's F.cross_entropy instead of taxoexpan_amd.loss.info_nce_loss -- each alone, same resident batches, same step count
    ab = {}
    if args.workload == "pgat" and world == 1:
        def timed(opt_, loss_fn=None, mdl=None):
            mdl = model if mdl is None else mdl
            for i in range(min(args.warmup, 5)):
                train_step(mdl, opt_, batches[i % len(batches)], target, world, loss_fn)
            ts, per = [], max(args.steps // 5, 1)        # median of five groups: a one-off host stall does not decide an A/B leg
            for _ in range(5):
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for i in range(per):
                    train_step(mdl, opt_, batches[i % len(batches)], target, world, loss_fn)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0_) / per)
            return float(np.median(ts))
        from taxoexpan_amd import ops as _ops_ab
        _ops_ab._NO_MATCH_FOLD = True                    # the graph vector hg = Z W^T formed, the matcher on 4,096 rows of it (DESIGN 4.9 off)
        ab["step_no_match_fold_ms"] = timed(opt)
        _ops_ab._NO_MATCH_FOLD = False
        _ops_ab._NO_SPLIT_GEMM = True                    # the first layer's projection and weight gradient on the fp32 MFMA (what rounds 1-5
        ab["step_fp32_mfma_ms"] = timed(opt)             # timed) instead of the bf16 pipe's six plane products
        _ops_ab._NO_SPLIT_GEMM = False
        ab["step_reference_model_py_ms"] = timed(opt, mdl=ReferenceCaller(model))
        ab["step_reference_model_py_routes"] = {k: _ops_ab.ROUTES.get(k) for k in ("match", "stack", "fold", "stack_bwd")}
        torch.autograd.set_multithreading_enabled(True)
        ab["step_default_autograd_ms"] = timed(opt)
        torch.autograd.set_multithreading_enabled(False)
        ab["step_torch_adam_ms"] = timed(torch.optim.Adam(model.parameters(), lr=LR, weight_decay=0, amsgrad=True))
        ab["step_torch_loss_ms"] = timed(opt, lambda out, tgt: F.cross_entropy(out, tgt, reduction="sum"))
        torch.autograd.set_multithreading_enabled(True)
        ab["step_reference_caller_ms"] = timed(torch.optim.Adam(model.parameters(), lr=LR, weight_decay=0, amsgrad=True),
                                               lambda out, tgt: F.cross_entropy(out, tgt, reduction="sum"))     # all three as train.py has them
        torch.autograd.set_multithreading_enabled(False)
        # ... and with ONE key added to the reference's config JSON ("optimizer": {"args": {..., "fused": true}}, train.py builds the
        # optimizer from it): torch's own single-launch Adam
        try:
            ab["step_torch_adam_fused_ms"] = timed(torch.optim.Adam(model.parameters(), lr=LR, weight_decay=0, amsgrad=True, fused=True))
            torch.autograd.set_multithreading_enabled(True)
            ab["step_reference_caller_fused_adam_ms"] = timed(torch.optim.Adam(model.parameters(), lr=LR, weight_decay=0, amsgrad=True, fused=True),
                                                              lambda out, tgt: F.cross_entropy(out, tgt, reduction="sum"))
        except (RuntimeError, TypeError, ValueError) as e:       # (a torch build without the fused implementation)
            ab["step_torch_adam_fused_ms"] = None
            ab["step_torch_adam_fused_error"] = str(e)[:200]
        torch.autograd.set_multithreading_enabled(False)

    # the same step with a NEW batch built inside it (what an epoch of train.py pays per step): not `value` -- the contract times the
    # hot path on resident inputs -- but reported next to it
    from taxoexpan_amd import graph as Gr
    dtax = Gr.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, device)
    n_fresh = min(args.steps, 20)
    build_stream = torch.cuda.Stream(device=device)      # the batch of step i+1 is built while step i runs (data_loaders.DeviceBatchLoader)
    torch.cuda.synchronize()                             # (the resident taxonomy is complete before the side stream reads it)
    for i in range(3):
        train_step(model, opt, fresh_batch(tax, dtax, 5000 + i, device, build_stream), target, world)
    torch.cuda.synchronize()
    from taxoexpan_amd.data_loaders import finish_device_batch

    def fresh_loop(repeated):
        for i in range(2):
            train_step(model, opt, finish_device_batch(fresh_batch_begin(tax, dtax, 5500 + i, device, build_stream, repeated), dtax.features), target, world)
        torch.cuda.synchronize()
        t0f = time.perf_counter()
        n_edges = 0
        pend = fresh_batch_begin(tax, dtax, 6000 + rank, device, build_stream, repeated)
        for i in range(n_fresh):                             # DeviceBatchLoader's schedule: batch i+1 is begun (sampled, uploaded, node
            b = finish_device_batch(pend, dtax.features)     # counts launched) before step i is enqueued, finished after it
            if i + 1 < n_fresh:
                pend = fresh_batch_begin(tax, dtax, 6000 + 17 * (i + 1) + rank, device, build_stream, repeated)
            train_step(model, opt, b, target, world)
            n_edges += b["n_edges"]
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0f) / max(n_fresh, 1), n_edges
    fresh_ms, fresh_edges = fresh_loop(False)                # query features stacked per pair, as the reference's collate hands them over
    fresh_rq_ms, _ = fresh_loop(True)                        # ... as ops.RepeatedRows (DeviceBatchLoader's default): one row per query
    tb0 = time.perf_counter()
    for i in range(n_fresh):
        fresh_batch(tax, dtax, 9000 + i, device)
    torch.cuda.synchronize()
    build_ms = 1e3 * (time.perf_counter() - tb0) / max(n_fresh, 1)

    assert_finite(model, train_step(model, opt, batches[0], target, world), "after the A/B and fresh-batch legs")
    roof_all, cpu, extra = None, None, None
    if rank == 0:
        # instrumented steps (HIP events around every launch, on the launch stream) at the clocks of the timed region: a few plain
        # steps first, then three passes over the resident batches
        for i in range(8):
            train_step(model, opt, batches[i % len(batches)], target, 1)      # (rank-local, like profile_step: no collectives)
        prof_batches = [batches[i % len(batches)] for i in range(3 * len(batches))]
        recs = [profile_step(model, opt, b, target) for b in prof_batches]
        roof_all = summarize_profile(recs, [b["n_edges"] for b in prof_batches], [b["n_nodes"] for b in prof_batches], workload=args.workload)
    if rank == 0 and world == 1 and args.workload == "pgat":
        tax_full, full_batches, hg_cs, q_cs = None, None, None, None
        if not (args.no_extra and args.no_cpu_baseline):
            tax_full = syn.make_named_taxonomy("mag_full", seed=47)
            full_batches = build_batches(tax_full, 2, seed0=7000, device=device)        # MAG-Full-shaped egonet batches (4,096 each)
        if not args.no_extra:
            extra, hg_cs, q_cs = extra_metrics(model, tax, device, batches, full_batches)
            for name, fn in (("mag_full", lambda: extra_metrics_mag_full(model, device, tax_full)),
                             ("step_32768_egonets", lambda: large_batch_step(tax, device, 8)),
                             ("step_pgcn", lambda: variant_step("pgcn", tax, device)),
                             ("step_pgat2", lambda: variant_step("pgat2", tax_full, device)),
                             ("step_semeval", lambda: semeval_step(device))):
                try:                                     # never let a secondary metric take the bench line down
                    extra[name] = fn()
                except Exception as exc:                 # noqa: BLE001
                    extra[name] = {"error": repr(exc)[:300]}
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(batches[0], model.state_dict(), full_batches[0] if full_batches else None, hg_cs, q_cs)
            if extra is not None:                        # the north star's ratios, same process, same batches
                if "fwd" in cpu and "pgat_fwd_eval_mag_full_batches_edges_per_s" in extra:
                    extra["gpu_over_cpu_pgat_fwd_mag_full_batches"] = extra["pgat_fwd_eval_mag_full_batches_edges_per_s"] / cpu["fwd"]["value"]
                if "scoring" in cpu:
                    extra["gpu_over_cpu_candidates_scored_literal_loop"] = extra["candidates_scored_per_s"] / cpu["scoring"]["value"]
                    extra["gpu_over_cpu_candidates_scored_factored"] = extra["candidates_scored_per_s"] / cpu["scoring"]["factored_value"]
    if world > 1:
        dist.barrier()
        if args.workload == "pgat" and not args.no_extra:
            try:                                     # never let the secondary metric take the bench line down
                extra = extra_metrics_sharded(model, device, world, rank)
            except Exception as exc:                 # noqa: BLE001
                extra = {"error": repr(exc)[:300]}
            dist.barrier()
            try:                                     # BASELINE configs[3]: the 2-layer MAG-Full step, data-parallel over the N ranks
                from taxoexpan_amd import synthetic as _syn
                extra["step_pgat2_dp"] = dp_step_pgat2(device, world, rank, _syn.make_named_taxonomy("mag_full", seed=47))
            except Exception as exc:                 # noqa: BLE001
                extra["step_pgat2_dp"] = {"error": repr(exc)[:300]}
            dist.barrier()

    if rank == 0:
        # the dominant kernel of the critical path: second-stream launches are listed in roofline_all with their (stretched) durations
        dom = next((r for r in roof_all if r["stream"] == "main"), roof_all[0])
        hbm = [r for r in roof_all if r["bound"] == "hbm"]
        copy_bw, copy_bw_torch = hbm_copy_ceiling(device)
        for r in hbm:                                # (beside the fraction of the 8 TB/s spec)
            r["frac_of_copy_ceiling"] = r["achieved"] * 1e9 / copy_bw
        def by_kernel(prefix):
            return next((r for r in roof_all if r["kernel"].startswith(prefix)), None)
        hb = hbm[0] if hbm else None
        pair = concurrent_pair(recs, dom["kernel"]) if dom["bound"] == "mfma" else None
        roofline = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
                    "frac": dom["frac"], "traffic": dom["traffic"], "traffic_source": dom["traffic_source"],
                    "kernel": dom["kernel"], "avg_us": dom["avg_us"], "flops": "algorithmic (unpadded operands)",
                    **({"pipe": dom["pipe"], "frac_of_f32_mfma_peak": dom["frac_of_f32_mfma_peak"]} if "pipe" in dom else {}),
                    "stream": dom["stream"], "launches_profiled": dom["launches"], "work_per_launch": dom["work_per_launch"],
                    # the HBM side as flat scalars (the north star's target is an HBM-utilisation one): the longest HBM-bound kernel of
                    # the step, algorithmic bytes / live launch duration against the 8 TB/s spec and against this box's copy rate
                    "hbm_kernel": hb["kernel"] if hb else None, "hbm_avg_us": hb["avg_us"] if hb else None,
                    "hbm_achieved_gbs": hb["achieved"] if hb else None, "hbm_frac": hb["frac"] if hb else None,
                    "hbm_frac_of_copy_ceiling": hb["frac_of_copy_ceiling"] if hb else None,
                    "hbm_traffic_ratio": (hb["traffic"] / hb["work_per_launch"] if hb and hb.get("traffic") else None),
                    "copy_ceiling_gbs": copy_bw / 1e9, "copy_ceiling_torch_gbs": copy_bw_torch / 1e9,
                    "pair_frac": pair["pair_frac"] if pair else None, "pair_kernel": pair["kernel"] if pair else None}
        for short, prefix in (("aggregate_fwd", "gat_aggregate_"), ("fused_bwd", "gat_fused_bwd"),
                              ("dx_pos", "gat_dx_pos_kernel"), ("bwd_dot", "cl_bwd_dot"), ("zsum", "cl_zsum")):
            r = by_kernel(prefix)
            if r is not None:
                roofline[f"hbm_{short}_frac"] = r["frac"]
                roofline[f"hbm_{short}_avg_us"] = r["avg_us"]
        mf = [r for r in roof_all if r["bound"] == "mfma" and r["stream"] == "main"]
        if mf:        # all main-stream MFMA launches of a step together: algorithmic flops / their summed durations
            # (each kernel's flops against ITS pipe's roof: time at the roof / time spent)
            roofline["mfma_main_stream_frac"] = sum(r["work_per_launch"] * r["launches"] / (r["peak"] * 1e12) for r in mf) / sum(r["total_us"] * 1e-6 for r in mf)
        if cpu is not None:                              # (flat copies: nested objects do not survive every consumer of this line)
            for leg in ("fwd", "scoring"):
                if leg in cpu:
                    cpu[f"{leg}_value"] = cpu[leg]["value"]
                    cpu[f"{leg}_unit"] = cpu[leg]["unit"]
            if "scoring" in cpu:
                cpu["scoring_factored_value"] = cpu["scoring"]["factored_value"]
        line = {
            "metric": "egonet_edges_per_sec_%s_fwd_bwd" % args.workload, "value": edges / elapsed, "unit": "egonet-edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "matrix_pipe": "bf16x3-split(6 products, fp32 accumulate)" if args.workload != "pgcn" else "fp32-mfma",
            "config": {"workload": WORKLOAD_TEXT[args.workload] + STEP_TEXT,
                       "output_layer": "folded behind the weighted-mean readout (exact re-association, DESIGN 4.1): G graph rows instead of N node rows"
                                       + ("" if args.workload == "pgcn" else "; and, the query rows of a training batch repeating 32 times, through the "
                                          "bilinear matcher (DESIGN 4.9): its three G x D x Kp products run on the 128 query runs -- every output and "
                                          "gradient still produced, all work inside the timed region"),
                       "egonets_per_step_per_gpu": N_QUERIES * (1 + NEG), "avg_edges_per_step_per_gpu": edges / args.steps / world,
                       "matrix_pipe": "the first layer's projection and weight gradient (and, in the eval extras, the scoring loop and the encoder's two large "
                                      "products) run on the bf16 MFMA as SIX exact plane products per fp32 product with fp32 accumulation -- every fp32 "
                                      "operand is the exact sum of three bf16 numbers; results are as close to float64 as an fp32 GEMM's "
                                      "(DESIGN 4.10, tests/test_gpu_split_gemm.py, tests/test_split_arithmetic.py); step_fp32_mfma_ms = the same "
                                      "step with those products on the fp32 MFMA; roofline_all prices them against 2.5 PF/s / 6",
                       "settle_steps": SETTLE_STEPS, "sanity": sanity, "routes": routes_timed, "lr": LR, "final_loss": final_loss,
                       "parallelism": f"dp{world}"},
            "roofline_all": roof_all,
            "extra": extra,
            "cpu_baseline": cpu,
            "roofline": roofline,
            # flat scalars, last so that they end the line: a fresh batch built inside every step (device egonet builder, one host sync),
            # and the secondary halves of BASELINE.json's metric
            "step_incl_batch_build_ms": fresh_ms, "batch_build_ms": build_ms, "step_incl_batch_build_repeated_queries_ms": fresh_rq_ms,
            "step_repeated_queries_ms": rq_ms, **ab,
            "egonet_edges_per_s_incl_batch_build": world * fresh_edges / max(n_fresh, 1) / (fresh_ms * 1e-3),
        }
        if world > 1:
            line["rccl_world"] = dist.get_world_size()
            line["collective_backend"] = dist.get_backend()          # "nccl" = RCCL; "gloo" only in the one-GPU test of this path
        if extra and world > 1:
            for k_out, path in (("step_pgat2_dp_ms", ("step_pgat2_dp", "ms_per_step")),
                                ("step_pgat2_dp_edges_per_s", ("step_pgat2_dp", "egonet_edges_per_s")),
                                ("candidates_scored_per_s_allgather", ("candidates_scored_per_s_allgather",)),
                                ("candidates_scored_per_s_fused_allreduce", ("candidates_scored_per_s_fused_allreduce",)),
                                ("allgather_gbs_per_rank", ("allgather_gbs_per_rank",)),
                                ("allreduce_counts_queries_per_s", ("allreduce_counts_queries_per_s",))):
                v = extra
                for k in path:
                    v = v.get(k) if isinstance(v, dict) else None
                if v is not None:
                    line[k_out] = v
        if extra:
            for k_out, path in (("pgat_fwd_eval_edges_per_s", ("pgat_fwd_eval_edges_per_s",)),
                                ("pgat_fwd_eval_mag_full_edges_per_s", ("pgat_fwd_eval_mag_full_batches_edges_per_s",)),
                                ("gpu_over_cpu_pgat_fwd_mag_full", ("gpu_over_cpu_pgat_fwd_mag_full_batches",)),
                                ("candidates_scored_per_s_mag_cs", ("candidates_scored_per_s",)),
                                ("candidates_scored_per_s_mag_full", ("mag_full", "candidates_scored_per_s")),
                                ("candidates_scored_per_s_mag_full_fused_rank", ("mag_full", "candidates_scored_per_s_fused_rank")),
                                ("mag_full_encode_edges_per_s", ("mag_full", "encode_edges_per_s")),
                                ("infer_top5_queries_per_s", ("mag_full", "infer_top5_queries_per_s")),
                                ("infer_top5_fused_over_composite", ("mag_full", "infer_top5_fused_over_composite")),
                                ("step_pgcn_ms", ("step_pgcn", "ms_per_step")), ("step_pgat2_ms", ("step_pgat2", "ms_per_step")),
                                ("step_semeval_ms", ("step_semeval", "ms_per_step")),
                                ("step_semeval_edges_per_s", ("step_semeval", "egonet_edges_per_s")),
                                ("step_32768_egonets_ms", ("step_32768_egonets", "ms_per_step")),
                                ("step_32768_egonets_edges_per_s", ("step_32768_egonets", "egonet_edges_per_s")),
                                ("step_32768_egonets_roofline_frac", ("step_32768_egonets", "roofline", "frac"))):
                v = extra
                for k in path:
                    v = v.get(k) if isinstance(v, dict) else None
                if v is not None:
                    line[k_out] = v
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
