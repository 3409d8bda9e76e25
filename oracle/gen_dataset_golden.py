#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden traces of the reference's data path from the UNMODIFIED
/root/reference/data_loader/dataset.py: raw `.terms/.taxo/.terms.embed` loading, the seeded train/validation/test split,
MaskedGraphDataset's node lists / parents / masks, the negative sampler and the egonets `__getitem__` emits.

Runs ONLY in the build container.  dataset.py is imported as it lies, against oracle/dgl_shim and oracle/gensim_shim
(both third-party dependencies are absent; see their headers -- parity-unpinned for what they restate).  It writes
  tests/golden/toy_taxo/toy.{terms,taxo,terms.embed}     a 150-term synthetic taxonomy in the README.md:21-51 formats (data)
  tests/golden/dataset_trace.json                         what the reference produced from those files

    python oracle/gen_dataset_golden.py
"""
import importlib.util
import json
import os
import random
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "dgl_shim"))
sys.path.insert(0, os.path.join(HERE, "gensim_shim"))

import numpy as np  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
TOY = os.path.join(OUT, "toy_taxo")
N_TERMS, DIM = 150, 8


def write_toy():
    """deterministic raw files: string ids, names with spaces, file order != topological order, a hub with 11 children,
    multi-parent nodes, several roots, duplicated edge lines, embedding rows in another order"""
    rs = np.random.RandomState(20)
    os.makedirs(TOY, exist_ok=True)
    topo = rs.permutation(N_TERMS)                      # topo[r] = term index of topological rank r
    edges = []
    hub = topo[3]
    for r in range(4, N_TERMS):
        c = topo[r]
        n_par = 1 + (rs.rand() < 0.25) + (rs.rand() < 0.05)
        if r < 15:
            pars = {hub}
        else:
            w = 1.0 / (1.0 + np.arange(r)) ** 0.8       # favour early (shallow) nodes -> heavy-tailed out-degree
            pars = set(topo[rs.choice(r, size=n_par, replace=False, p=w / w.sum())].tolist())
        for p in pars:
            edges.append((p, c))
    edges = [edges[i] for i in rs.permutation(len(edges))]
    edges += edges[:5]                                   # duplicate lines: a DiGraph keeps one edge
    tx = [f"T{1000 + 7 * i}" for i in range(N_TERMS)]
    with open(os.path.join(TOY, "toy.terms"), "w") as f:
        for i in range(N_TERMS):
            f.write(f"{tx[i]}\tterm number {i}\n")
        f.write("\n")
    with open(os.path.join(TOY, "toy.taxo"), "w") as f:
        for p, c in edges:
            f.write(f"{tx[p]}\t{tx[c]}\n")
    emb = rs.randn(N_TERMS, DIM).astype(np.float32)
    with open(os.path.join(TOY, "toy.terms.embed"), "w") as f:
        f.write(f"{N_TERMS} {DIM}\n")
        for i in rs.permutation(N_TERMS):
            f.write(tx[i] + " " + " ".join(f"{v:.6f}" for v in emb[i]) + "\n")


def egonet_record(g):
    assert g.number_of_edges() == 2 * g.number_of_nodes() - 1
    return [g.ndata["_id"].tolist(), g.ndata["pos"].tolist()]


def main():
    write_toy()
    spec = importlib.util.spec_from_file_location("ref_dataset", os.path.join(REF, "data_loader", "dataset.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    tmp = tempfile.mkdtemp(dir=OUT)
    try:
        for fn in os.listdir(TOY):
            shutil.copy(os.path.join(TOY, fn), tmp)
        raw = ref.MAGDataset(name="toy", path=tmp, raw=True)            # dataset.py:82-194 (also writes its pickle into tmp)
        out = dict(vocab=raw.vocab, train=list(raw.train_node_ids), validation=list(raw.validation_node_ids),
                   test=list(raw.test_node_ids), features=raw.g_full.ndata["x"].numpy().round(6).tolist(),
                   full_edges=[raw.g_full.edges()[0].tolist(), raw.g_full.edges()[1].tolist()], modes={})
        runs = {
            "train": dict(mode="train", sampling_mode=1, negative_size=6, expand_factor=4, cache_refresh_time=3, normalize_embed=True),
            "validation": dict(mode="validation", sampling_mode=0, negative_size=6, expand_factor=4, cache_refresh_time=3,
                               normalize_embed=True),
            "test": dict(mode="test", sampling_mode=0, negative_size=6, expand_factor=4, cache_refresh_time=3, normalize_embed=False),
            "test_topk": dict(mode="test", sampling_mode=0, negative_size=6, expand_factor=4, cache_refresh_time=3,
                              normalize_embed=True, test_topk=5),
        }
        for name, kw in runs.items():
            ds = ref.MaskedGraphDataset(raw, **kw)                       # dataset.py:208-283
            rec = dict(args=kw, node_list=list(ds.node_list), node2parents={str(k): v for k, v in ds.node2parents.items()},
                       node2masks={str(k): sorted(v) for k, v in ds.node2masks.items()}, all_positions=sorted(ds.all_positions),
                       graph_edges=sorted([list(e) for e in ds.graph.edges()]), node_features=ds.node_features.numpy().round(6).tolist())
            random.seed(1234)
            trace = []
            order = list(range(len(ds))) * (2 if name == "train" else 1)
            if name.startswith("test"):
                order = order[:6]
            for idx in order:                                            # dataset.py:293-332
                inst = ds[idx]
                q = ds.node_list[idx]
                for (g, qf, lab) in inst:
                    assert np.array_equal(qf.numpy(), ds.node_features[q].numpy())
                trace.append(dict(query=q, labels=[t[2] for t in inst], egonets=[egonet_record(t[0]) for t in inst]))
            rec["trace"] = trace
            out["modes"][name] = rec
    finally:
        shutil.rmtree(tmp)
    with open(os.path.join(OUT, "dataset_trace.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", os.path.join(OUT, "dataset_trace.json"), os.path.getsize(os.path.join(OUT, "dataset_trace.json")), "bytes")


if __name__ == "__main__":
    main()
