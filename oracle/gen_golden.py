#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz from the UNMODIFIED reference.

Runs ONLY in the build container (needs /root/reference).  It puts oracle/dgl_shim (a pure-torch
stand-in for the un-installable DGL 0.4) and /root/reference on sys.path, imports the reference's
own model/model.py (TaxoExpan -> model_zoo.py), model/metric.py and model/loss.py, drives them exactly
like trainer/trainer.py:45-60 and test_fast.py:116-133 do, and stores what they produce.
No reference source is copied: the fixtures hold arrays (inputs via seeds, outputs, gradients).

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "dgl_shim"))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
sys.modules.setdefault("ipdb", types.ModuleType("ipdb"))  # model/model.py:8 imports it, never uses it

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dgl  # noqa: E402  (the shim)
import model.loss as ref_loss  # noqa: E402
import model.metric as ref_metric  # noqa: E402
import model.model as ref_model  # noqa: E402
import model.model_zoo as ref_zoo  # noqa: E402

import golden_cases as gc  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def build_egonet(k, m, x_rows):
    """The DGL calls of dataset.py:429-435, verbatim in meaning."""
    n = k + 1 + m
    g = dgl.DGLGraph()
    g.add_nodes(n, {"x": x_rows, "_id": torch.arange(n), "pos": torch.tensor([0] * k + [1] + [2] * m)})
    g.add_edges(list(range(k)), k)
    g.add_edges(k, list(range(k + 1, n)))
    g.add_edges(g.nodes(), g.nodes())
    return g


class FixedMask(torch.nn.Module):
    """stands in for nn.Dropout in the reference run: a FIXED keep mask and the 1/(1-p) scale."""

    def __init__(self, keep, p):
        super().__init__()
        self.keep, self.scale = torch.from_numpy(keep), 1.0 / (1.0 - p)

    def forward(self, x):
        return x * self.keep.reshape(x.shape) * self.scale


def run_case(name, spec):
    torch.manual_seed(0)
    shapes, x, q = gc.make_inputs(spec)
    params = gc.make_params(spec)
    drop = spec.get("dropout")
    opts = dict(in_dim=spec["in_dim"], hidden_dim=spec["hidden_dim"], out_dim=spec["out_dim"], pos_dim=spec["pos_dim"],
                num_layers=spec["num_layers"], heads=spec["heads"], feat_drop=(drop[0] if drop else 0.1),
                attn_drop=(drop[1] if drop else 0.1), hidden_drop=(drop[0] if drop else 0.1),
                out_drop=(drop[0] if drop else 0.1))
    model = ref_model.TaxoExpan(spec["prop"], spec["readout"], spec["match"], **opts)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)  # pins names+shapes

    graphs, off = [], 0
    xt = torch.from_numpy(x)
    for (k, m) in shapes:
        n = k + 1 + m
        graphs.append(build_egonet(k, m, xt[off:off + n]))
        off += n
    bg = dgl.batch(graphs)
    n_nodes, n_edges = bg.number_of_nodes(), bg.number_of_edges()

    is_gat = spec["prop"] in ("PGAT", "GAT")
    layers = model.graph_propagate.gat_layers if is_gat else model.graph_propagate.layers
    save = {}
    if drop:
        model.train()
        in_dims = [(l.fc.weight.shape[1] if is_gat else l.weight.shape[0]) for l in layers]
        masks = gc.make_dropout_masks(spec, in_dims, n_nodes, n_edges, spec["heads"] if is_gat else None)
        for l, (layer, (fk, ak)) in enumerate(zip(layers, masks)):
            if is_gat:
                layer.feat_drop = FixedMask(fk, drop[0])
                layer.attn_drop = FixedMask(ak, drop[1]) if drop[1] > 0 else torch.nn.Identity()
            else:
                layer.dropout = FixedMask(fk, drop[0])
    else:
        model.eval()

    caps = {}

    def layer_hook(idx):
        def hook(mod, inp, outp):
            caps[f"layer{idx}_out"] = outp.detach().clone()
            if is_gat:
                caps[f"layer{idx}_alpha"] = inp[0].edata["a_drop"].detach().clone()
        return hook
    for i, layer in enumerate(layers):
        layer.register_forward_hook(layer_hook(i))
    model.readout.register_forward_hook(lambda mod, inp, outp: caps.__setitem__("hg", outp.detach().clone()))

    # trainer.py:45-60
    nf = torch.from_numpy(q)
    h = bg.ndata.pop("x")
    prediction = model(bg, h, nf)
    n_q = spec["n_queries"]
    loss = ref_loss.info_nce_loss(prediction.reshape(n_q, -1), torch.zeros(n_q, dtype=torch.long))
    loss.backward()

    save["src"] = bg._src.numpy().astype(np.int64)
    save["dst"] = bg._dst.numpy().astype(np.int64)
    save["batch_num_nodes"] = np.asarray(bg.batch_num_nodes, dtype=np.int64)
    save["scores"] = prediction.detach().numpy()
    save["loss"] = np.asarray(loss.item(), dtype=np.float64)
    save["hn"] = bg.ndata["h"].detach().numpy()[::gc.row_steps(spec)[0]].copy()
    for k, v in caps.items():
        # large-dimension cases keep every 5th (slim: 40th) node row of the per-layer outputs (fixture size)
        save[k] = v.numpy()[::gc.row_steps(spec)[1]].copy() if k.endswith("_out") else v.numpy()
    for k, p in model.named_parameters():
        g = p.grad.detach().numpy()
        if spec["full"] or g.size <= 4096:
            save["grad:" + k] = g
        else:
            flat = g.reshape(-1)
            stride = max(1, flat.size // 2048)
            save["gradsample:" + k] = flat[::stride].copy()
            save["gradstats:" + k] = np.asarray([flat.astype(np.float64).sum(), np.abs(flat.astype(np.float64)).sum(),
                                                 stride], dtype=np.float64)
    if spec["full"]:
        save["x"], save["q"] = x, q
        for k, v in params.items():
            save["param:" + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)
    print(f"{name}: G={len(shapes)} N={n_nodes} E={n_edges} loss={loss.item():.6f}")


def run_scoring():
    """test_fast.py:116-133 per-query loop with the reference's LBM/BIM + metric.py ranks."""
    hg, qs, W, positives = gc.make_scoring_inputs()
    save = {}
    for kind, cls in (("lbm", ref_zoo.LBM), ("bim", ref_zoo.BIM)):
        mod = cls(hg.shape[1], qs.shape[1])
        mod.load_state_dict({"W.weight": torch.from_numpy(W)})
        S, R = [], []
        cand = np.arange(hg.shape[0])
        with torch.no_grad():
            for qi in range(qs.shape[0]):
                nf = torch.from_numpy(qs[qi])
                energy = mod(torch.from_numpy(hg), nf.expand(hg.shape[0], -1))          # test_fast.py:122-123
                S.append(energy.squeeze(1).numpy().copy())
                tmp = np.isin(cand, positives[qi])                                          # rearrange, :16-22
                correct, incorrect = np.where(tmp)[0], np.where(~tmp)[0]
                labels = torch.cat((torch.ones(len(correct)), torch.zeros(len(incorrect)))).int()
                es = torch.cat((energy[correct, :], energy[incorrect, :]))
                ranks = ref_metric.obtain_ranks(es, labels, mode=1)                       # :133, info_nce => mode 1
                R.append(np.asarray(ranks[0], dtype=np.int64))
        save[f"S_{kind}"] = np.stack(S)
        save[f"ranks_{kind}"] = np.concatenate(R)
        save[f"ranks_{kind}_off"] = np.cumsum([0] + [len(r) for r in R]).astype(np.int64)
        # metrics of metric.py on those ranks
        save[f"metrics_{kind}"] = np.asarray([
            np.mean([ref_metric.macro_mr([r.tolist()]) for r in R]),
            np.mean([ref_metric.hit_at_1([r.tolist()]) for r in R]),
            np.mean([ref_metric.mrr_scaled_10([r.tolist()]) for r in R])], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "scoring.npz"), **save)
    print("scoring: S", save["S_lbm"].shape)


def run_newterms():
    """infer.py:23-38 + :96-106 on new-term vectors of realistic magnitude: rows divided by their SUM (float64 numpy, then the float32
    of gensim's KeyedVectors and of `torch.tensor(kv[q], dtype=torch.float32)`), the reference's LBM / BIM on every candidate, and
    the top-5 of `sorted(enumerate(scores), key=...)[:5]` -- descending (info_nce losses) and ascending (the others).  LBM's exp
    overflows to inf on the rows with a tiny sum: the order of equal infs is Python's stable sort, i.e. candidate order."""
    hg, raw, W = gc.make_newterm_inputs()
    nf = np.array(raw)
    row_sums = nf.sum(axis=1)                                                       # infer.py:33-35
    nf = nf / row_sums[:, np.newaxis]
    nf32 = np.asarray(nf, dtype=np.float32)                                         # KeyedVectors.add -> REAL
    save = {"nf32": nf32}
    for kind, cls in (("lbm", ref_zoo.LBM), ("bim", ref_zoo.BIM)):
        mod = cls(hg.shape[1], nf32.shape[1])
        mod.load_state_dict({"W.weight": torch.from_numpy(W)})
        S, top_d, top_a = [], [], []
        with torch.no_grad():
            for qi in range(nf32.shape[0]):
                q = torch.tensor(nf32[qi], dtype=torch.float32)
                energy = mod(torch.from_numpy(hg), q.expand(hg.shape[0], -1))           # infer.py:96-98
                scores = energy.cpu().squeeze_().tolist()
                S.append(np.asarray(scores, dtype=np.float32))
                top_d.append([e[0] for e in sorted(enumerate(scores), key=lambda x: -x[1])[:5]])   # infer.py:101
                top_a.append([e[0] for e in sorted(enumerate(scores), key=lambda x: x[1])[:5]])    # infer.py:103
        save[f"S_{kind}"] = np.stack(S)
        save[f"top5_desc_{kind}"] = np.asarray(top_d, dtype=np.int64)
        save[f"top5_asc_{kind}"] = np.asarray(top_a, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "newterms.npz"), **save)
    n_inf = int(np.isinf(save["S_lbm"]).sum())
    print(f"newterms: S {save['S_lbm'].shape}, {n_inf} inf scores, max |q| {np.abs(nf32).max():.1f}")


def run_extras():
    """modules of model_zoo.py that model/model.py never instantiates: the NTN matcher (:331-346) and GATLayer's residual
    branch (:98-103, with res_fc and with the broadcast identity), forward + gradients of sum(out * coef)."""
    rs = np.random.RandomState(77)
    f32 = lambda *shape: (rs.randn(*shape) * 0.5).astype(np.float32)
    save = {}
    # --- NTN
    ntn = ref_zoo.NTN(12, 7, k=4)
    P = {"u_R.weight": f32(1, 4), "W.weight": f32(4, 12, 7), "W.bias": f32(4), "V.weight": f32(4, 19)}
    ntn.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}, strict=True)
    e1, e2, coef = f32(9, 12), f32(9, 7), f32(9, 1)
    t1, t2 = torch.from_numpy(e1).requires_grad_(), torch.from_numpy(e2).requires_grad_()
    out = ntn(t1, t2)
    (out * torch.from_numpy(coef)).sum().backward()
    save.update({"ntn.e1": e1, "ntn.e2": e2, "ntn.coef": coef, "ntn.out": out.detach().numpy(), "ntn.d_e1": t1.grad.numpy(),
                 "ntn.d_e2": t2.grad.numpy()})
    for k, v in P.items():
        save["ntn.p." + k] = v
    for k, v in ntn.named_parameters():
        save["ntn.g." + k] = v.grad.numpy()
    # --- residual GATLayer on the EDGE_SHAPES batch
    shapes = gc.EDGE_SHAPES
    n = sum(k + 1 + m for k, m in shapes)
    for tag, (din, dout, H) in {"res_fc": (10, 6, 3), "res_id": (6, 6, 2)}.items():
        layer = ref_zoo.GATLayer(din, dout, H, feat_drop=0.0, attn_drop=0.0, residual=True)
        P = {"fc.weight": f32(H * dout, din), "attn_l": f32(1, H, dout), "attn_r": f32(1, H, dout)}
        if din != dout:
            P["res_fc.weight"] = f32(H * dout, din)
        layer.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}, strict=True)
        x, coef = f32(n, din), f32(n, H, dout)
        tx = torch.from_numpy(x).requires_grad_()
        off, graphs = 0, []
        for (k, m) in shapes:
            graphs.append(build_egonet(k, m, tx[off:off + k + 1 + m]))
            off += k + 1 + m
        bg = dgl.batch(graphs)
        out = layer(bg, tx)
        (out * torch.from_numpy(coef)).sum().backward()
        save.update({f"{tag}.x": x, f"{tag}.coef": coef, f"{tag}.out": out.detach().numpy(), f"{tag}.d_x": tx.grad.numpy()})
        for k, v in P.items():
            save[f"{tag}.p.{k}"] = v
        for k, v in layer.named_parameters():
            save[f"{tag}.g.{k}"] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "extras.npz"), **save)
    print("extras:", len(save), "arrays")


def run_metrics():
    """model/metric.py:62-96 (numpy only) on a fixed heavy-tailed rank set -> tests/golden/metrics.json"""
    import json
    rs = np.random.RandomState(9)
    npos = rs.randint(1, 5, size=200)
    ranks = np.concatenate([np.floor(np.abs(rs.standard_cauchy(k)) * 7).astype(np.int64) + 1 for k in npos])
    off = np.concatenate([[0], np.cumsum(npos)])
    all_ranks = [ranks[off[i]:off[i + 1]].tolist() for i in range(len(npos))]
    out = dict(ranks=ranks.tolist(), pos_off=off.tolist())
    for name in ("macro_mr", "micro_mr", "hit_at_1", "hit_at_3", "hit_at_5", "mrr_scaled_10", "combined_metrics"):
        out[name] = float(getattr(ref_metric, name)(all_ranks))
    with open(os.path.join(OUT, "metrics.json"), "w") as f:
        json.dump(out, f)
    print("metrics:", {k: v for k, v in out.items() if k not in ("ranks", "pos_off")})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--extras-only" in sys.argv:
        run_extras()
        sys.exit(0)
    if "--newterms-only" in sys.argv:
        run_newterms()
        sys.exit(0)
    if "--metrics-only" in sys.argv:
        run_metrics()
        sys.exit(0)
    if "--case" in sys.argv:                    # one case only (adding a case does not rewrite the others)
        name = sys.argv[sys.argv.index("--case") + 1]
        run_case(name, gc.CASES[name])
        sys.exit(0)
    for name, spec in gc.CASES.items():
        run_case(name, spec)
    run_scoring()
    run_newterms()
    run_extras()
    run_metrics()
