"""CPU: host-side logic -- the DGL-0.4 graph surface, CSR construction, synthetic taxonomies / egonet batching,
module/state-dict compatibility with the reference (names + shapes from the golden specs)."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
import txe_oracle as orc
from golden_util import GOLDEN_DIR


def test_dgl_surface_and_batch_match_reference_layout():
    from taxoexpan_amd import graph as G
    shapes = gc.EDGE_SHAPES
    graphs = []
    for (k, m) in shapes:      # dataset.py:429-435 call sequence
        n = k + 1 + m
        g = G.DGLGraph()
        g.add_nodes(n, {"x": torch.randn(n, 4), "_id": torch.arange(n), "pos": torch.tensor([0] * k + [1] + [2] * m)})
        g.add_edges(list(range(k)), k)
        g.add_edges(k, list(range(k + 1, n)))
        g.add_edges(g.nodes(), g.nodes())
        assert g.number_of_nodes() == n and g.number_of_edges() == 2 * n - 1
        graphs.append(g)
    bg = G.batch(graphs)
    ref = orc.batch_egonets(shapes)
    assert bg.batch_size == len(shapes)
    assert bg.batch_num_nodes == [k + 1 + m for k, m in shapes]
    assert np.array_equal(bg._src, ref["src"].numpy()) and np.array_equal(bg._dst, ref["dst"].numpy())
    assert torch.equal(bg.ndata["pos"], ref["pos"])
    assert torch.equal(bg.in_degrees(), torch.bincount(ref["dst"], minlength=ref["num_nodes"]))
    x = bg.ndata.pop("x")
    assert x.shape[0] == bg.number_of_nodes() and "x" not in bg.ndata
    # vectorised constructor == per-egonet construction + dgl.batch
    vg = G.BatchedDGLGraph.from_egonet_shapes([s[0] for s in shapes], [s[1] for s in shapes])
    assert np.array_equal(vg._src, bg._src) and np.array_equal(vg._dst, bg._dst)
    assert torch.equal(vg.ndata["pos"], bg.ndata["pos"]) and vg.batch_num_edges == bg.batch_num_edges


def test_host_csr_views():
    from taxoexpan_amd import graph as G
    rs = np.random.RandomState(0)
    n, e = 50, 400
    src, dst = rs.randint(0, n, e), rs.randint(0, n, e)
    g = G.DGLGraph()
    g.add_nodes(n)
    g.add_edges(src, dst)
    c = g.csr("cpu")
    rp, col, eid = c.rowptr_in.numpy(), c.col_src.numpy(), c.eid_in.numpy()
    for v in range(n):
        seg = eid[rp[v]:rp[v + 1]]
        assert np.all(dst[seg] == v) and np.all(np.diff(seg) > 0)          # stable: edge-id order inside a segment
        assert np.array_equal(col[rp[v]:rp[v + 1]], src[seg])
    rpo, cd, po = c.rowptr_out.numpy(), c.col_dst.numpy(), c.pos_out.numpy()
    for u in range(n):
        for j in range(rpo[u], rpo[u + 1]):
            p = po[j]
            assert col[p] == u and dst[eid[p]] == cd[j]
    assert c.graph_off.tolist() == [0, n]
    with pytest.raises(ValueError):
        g.add_edges([0], [n])


def test_synthetic_taxonomy_and_egonets():
    from taxoexpan_amd import synthetic as syn
    tax = syn.make_taxonomy(3000, 4700, 16, seed=1)
    assert tax.par_ptr[-1] == tax.n_edges and tax.chd_ptr[-1] == tax.n_edges
    # DAG in topological order, distinct parents
    for v in range(1, 3000, 97):
        ps = tax.par_idx[tax.par_ptr[v]:tax.par_ptr[v + 1]]
        assert len(ps) >= 1 and np.all(ps < v) and len(set(ps.tolist())) == len(ps)
    leaf_frac = len(tax.leaves()) / tax.n_nodes
    assert 0.6 < leaf_frac < 0.95
    np.testing.assert_allclose(tax.features.norm(dim=1).numpy(), 1.0, rtol=1e-5)
    cand, val, test = syn.split_candidates(tax)
    assert len(val) == len(test) and len(cand) + len(val) + len(test) == tax.n_nodes
    g = syn.egonet_batch(tax, cand[:200], expand_factor=5, seed=3)
    n = np.asarray(g.batch_num_nodes)
    off = np.concatenate([[0], np.cumsum(n)])
    ids, pos = g.ndata["_id"].numpy(), g.ndata["pos"].numpy()
    for i, a in enumerate(cand[:200]):
        p = pos[off[i]:off[i + 1]]
        k, m = int((p == 0).sum()), int((p == 2).sum())
        assert ids[off[i] + k] == a and m <= 5
        assert sorted(ids[off[i]:off[i] + k].tolist()) == sorted(tax.par_idx[tax.par_ptr[a]:tax.par_ptr[a + 1]].tolist())
        assert set(ids[off[i] + k + 1:off[i + 1]].tolist()) <= set(tax.chd_idx[tax.chd_ptr[a]:tax.chd_ptr[a + 1]].tolist())
    assert torch.equal(g.ndata["x"], tax.features[g.ndata["_id"]])
    gb, qf, lab = syn.training_batch(tax, 8, 3, seed=2)
    assert gb.batch_size == 32 and qf.shape == (32, 16) and lab.reshape(8, 4)[:, 0].tolist() == [1] * 8
    # the positive egonet never contains its own query among the siblings (instance_mode 1, dataset.py:421-424)
    assert gb.number_of_edges() == 2 * gb.number_of_nodes() - gb.batch_size


@pytest.mark.parametrize("name", list(gc.CASES))
def test_state_dict_names_and_shapes_match_reference(name):
    """strict load of the reference-keyed parameter set (keys/shapes were pinned by load_state_dict(strict=True) on the
    reference's own TaxoExpan in oracle/gen_golden.py)"""
    from taxoexpan_amd import TaxoExpan
    spec = gc.CASES[name]
    params = gc.make_params(spec)
    model = TaxoExpan(spec["prop"], spec["readout"], spec["match"], in_dim=spec["in_dim"], hidden_dim=spec["hidden_dim"],
                      out_dim=spec["out_dim"], pos_dim=spec["pos_dim"], num_layers=spec["num_layers"], heads=spec["heads"],
                      feat_drop=0.1, attn_drop=0.1, hidden_drop=0.1, out_drop=0.1)
    sd = model.state_dict()
    assert list(sd.keys()) == list(params.keys())
    for k, v in params.items():
        assert tuple(sd[k].shape) == v.shape, k
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    assert hasattr(model, "graph_propagate") and hasattr(model, "readout") and hasattr(model, "match")


def test_unknown_method_strings_fall_through_silently():
    from taxoexpan_amd import TaxoExpan
    m = TaxoExpan("nope", "MR", "BIM", in_dim=4, hidden_dim=4, out_dim=4, pos_dim=2, num_layers=1, heads=[1, 1],
                  feat_drop=0.1, attn_drop=0.1, hidden_drop=0.1, out_drop=0.1)
    assert not hasattr(m, "graph_propagate")       # model/model.py:43 `assert "<str>"` never fires


def test_generic_dgl_message_passing_raises_loudly():
    from taxoexpan_amd import graph as G
    g = G.DGLGraph()
    g.add_nodes(2)
    with pytest.raises(NotImplementedError):
        g.update_all(None, None)


def test_metrics_match_reference_metric_py():
    """taxoexpan_amd.metric on the flat (ranks, pos_off) layout == the reference's model/metric.py on its nested lists
    (tests/golden/metrics.json: oracle/gen_golden.py --metrics-only, which imports the unmodified metric.py)"""
    import json
    import os
    from golden_util import GOLDEN_DIR
    from taxoexpan_amd import metric
    z = json.load(open(os.path.join(GOLDEN_DIR, "metrics.json")))
    ranks, off = torch.tensor(z["ranks"], dtype=torch.int32), torch.tensor(z["pos_off"])
    for name in ("macro_mr", "micro_mr", "hit_at_1", "hit_at_3", "hit_at_5", "mrr_scaled_10", "combined_metrics"):
        assert abs(getattr(metric, name)(ranks, off) - z[name]) <= 1e-12 * max(1.0, abs(z[name])), name
    lists = metric.as_rank_lists(ranks, off)
    assert len(lists) == len(z["pos_off"]) - 1 and sum(len(r) for r in lists) == len(z["ranks"])


def test_gathered_rows_is_an_ordinary_tensor_to_everyone_else():
    """ops.GatheredRows (node features kept as rows of the taxonomy table): shape / device / arithmetic / torch
    functions / indexing see table[index]; the projection cache nests and remembers the largest expected row count"""
    import torch
    from taxoexpan_amd import ops
    table = torch.arange(15, dtype=torch.float32).reshape(5, 3)
    idx = torch.tensor([4, 0, 0, 2], dtype=torch.int32)
    x = ops.GatheredRows(table, idx)
    want = table[idx.long()]
    assert tuple(x.shape) == (4, 3) and x.dim() == 2 and len(x) == 4 and x.device == table.device and x.dtype == torch.float32
    assert torch.equal(x.tensor(), want) and torch.equal(x + 1, want + 1) and torch.equal(2 * x, 2 * want)
    assert torch.equal(torch.cat([x, x], 0), torch.cat([want, want], 0)) and torch.equal(x[1], want[1]) and torch.equal(x.sum(1), want.sum(1))
    assert x.to(table.device) is x and "GatheredRows" in repr(x)
    assert ops._PROJ_CACHE is None
    with ops.projection_cache(10):
        assert ops._PROJ_CACHE["expected_rows"] == 10
        with ops.projection_cache(3):
            assert ops._PROJ_CACHE["expected_rows"] == 10
        assert ops._PROJ_CACHE is not None
    assert ops._PROJ_CACHE is None
    # the table path needs a GPU, no gradients and no dropout: on the host it is never chosen
    assert not ops._use_table(x, False, 0.0) and not ops._use_table(want, False, 0.0)


def test_repeated_rows_is_the_stacked_matrix_to_everyone_else():
    """ops.RepeatedRows (the query features of a training batch as one row per query + run offsets): the runs found from the ids,
    dense() = the reference collate's stack (data_loaders.py:9-28), the tensor-like surface the model code touches; BIM / LBM never
    take a CPU path with it; a matcher without a runs form sees the stacked tensor"""
    from taxoexpan_amd import ops
    from taxoexpan_amd.model_zoo import LBM
    table = torch.arange(40, dtype=torch.float32).reshape(10, 4)
    ids = np.array([3, 3, 3, 7, 1, 1, 3, 9, 9, 9, 9])
    rr = ops.RepeatedRows.from_ids(table, ids)
    assert rr.rows.tolist() == table[[3, 7, 1, 3, 9]].tolist() and rr.run_off.tolist() == [0, 3, 4, 6, 7, 11]
    assert rr.shape == (11, 4) and rr.dim() == 2 and rr.dtype == torch.float32 and rr.device == table.device and not rr.requires_grad
    assert torch.equal(rr.dense(), table[torch.from_numpy(ids)]) and torch.equal(ops.dense_rows(rr), rr.dense()) and ops.dense_rows(table) is table
    empty = ops.RepeatedRows.from_ids(table, np.zeros(0, dtype=np.int64))
    assert empty.shape == (0, 4) and empty.run_off.tolist() == [0] and empty.dense().shape == (0, 4)
    moved = rr.to("cpu")
    assert torch.equal(moved.dense(), rr.dense()) and moved.n_rows == 11
    with pytest.raises((RuntimeError, ValueError, AssertionError)):
        LBM(6, 4)(torch.randn(11, 6), rr)                     # host tensors: no CPU fallback for the runs form either


def test_loss_and_optimizer_have_no_cpu_path():
    """taxoexpan_amd.loss / optim mirror model/loss.py:52-57 and torch.optim.Adam's constructor, and fail loudly off the GPU"""
    import pytest
    import torch
    from taxoexpan_amd.loss import info_nce_loss
    from taxoexpan_amd.optim import Adam
    with pytest.raises(RuntimeError):
        info_nce_loss(torch.zeros(4, 3), torch.zeros(4, dtype=torch.long))
    with pytest.raises(ValueError):
        info_nce_loss(torch.zeros(4), None)
    p = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(ValueError):
        Adam([p], lr=-1.0)
    with pytest.raises(ValueError):
        Adam([p], betas=(0.9, 1.0))
    opt = Adam([p], lr=1e-3, amsgrad=True)
    assert opt.defaults == dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True)
    opt.step()                                   # no gradient yet: nothing to do, nothing launched
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        opt.step()


def _toy_dataset(tmp_path, mode="test", **kw):
    import shutil
    from taxoexpan_amd.dataset import MAGDataset, MaskedGraphDataset
    for fn in os.listdir(os.path.join(GOLDEN_DIR, "toy_taxo")):
        shutil.copy(os.path.join(GOLDEN_DIR, "toy_taxo", fn), tmp_path)
    return MaskedGraphDataset(MAGDataset("toy", str(tmp_path), raw=True), mode=mode, sampling_mode=0, **kw)


def test_masked_graph_dataset_kv_like_the_reference(tmp_path):
    """dataset.py:227-229: kv maps str(node id) -> (normalised) feature row; test_fast.py:121 reads `kv[str(query)]`"""
    ds = _toy_dataset(tmp_path, normalize_embed=True)
    x = ds.node_features.numpy()
    for q in list(ds.node_list)[:5] + [0, len(ds.vocab) - 1]:
        np.testing.assert_array_equal(ds.kv[str(q)], x[q])
        assert torch.equal(torch.tensor(ds.kv[str(q)], dtype=torch.float32), ds.node_features[q])
    assert ds.kv.vector_size == x.shape[1] and len(ds.kv) == len(ds.vocab) and "0" in ds.kv and "nope" not in ds.kv
    with pytest.raises(KeyError):
        ds.kv["nope"]
    pool = [str(i) for i in (3, 1, 7)]
    want = [1.0 - float(x[i] @ x[2]) / float(np.linalg.norm(x[i]) * np.linalg.norm(x[2])) for i in (3, 1, 7)]
    np.testing.assert_allclose(ds.kv.distances("2", pool), want, rtol=1e-6)
    assert list(ds.graph.nodes()) == list(ds.graph.nodes)                    # infer.py:80 calls it, our code reads the attribute


def test_topk_parents_is_pythons_stable_sort():
    """infer.py:100-106: sorted(enumerate(scores), key=-score)[:5] -- ties come out in candidate order"""
    from taxoexpan_amd.scoring import topk_parents
    torch.manual_seed(0)
    ids = torch.arange(100, 123)
    for larger in (True, False):
        S = torch.randint(0, 4, (40, 23)).float()
        S[3] = float("inf")
        S[4, :10] = float("inf")
        S[5] = torch.randn(23)
        got = topk_parents(S, ids, 5, larger)
        for q in range(S.shape[0]):
            key = (lambda e: -e[1]) if larger else (lambda e: e[1])
            assert got[q].tolist() == [int(ids[e[0]]) for e in sorted(enumerate(S[q].tolist()), key=key)[:5]]
    assert topk_parents(torch.randn(3, 2), torch.arange(2), 5).shape == (3, 2)
    # NaN scores rank last instead of selecting nothing (the row used to pick only filler columns: an out-of-range gather on the
    # GPU); -inf scores tie with the filler key and still come out in candidate order
    S = torch.tensor([[1.0, float("nan"), 3.0, 2.0, float("nan"), 0.5, 7.0],
                      [float("nan")] * 7,
                      [float("-inf"), 1.0, float("-inf"), float("-inf"), float("-inf"), float("-inf"), float("-inf")]])
    got = topk_parents(S, torch.arange(7), 5, True).tolist()
    assert got[0] == [6, 2, 3, 0, 5] and got[1] == [0, 1, 2, 3, 4] and got[2] == [1, 0, 2, 3, 4]
    assert topk_parents(S[:1], torch.arange(7), 5, False).tolist()[0] == [5, 0, 3, 2, 6]


def test_data_loader_uses_the_cache_after_the_first_raw_load(tmp_path, monkeypatch):
    """the raw text files are parsed once; later loaders (validation / test / other ranks) read <name>.txe.npz"""
    import shutil
    from taxoexpan_amd import data_loaders, dataset
    for fn in os.listdir(os.path.join(GOLDEN_DIR, "toy_taxo")):
        shutil.copy(os.path.join(GOLDEN_DIR, "toy_taxo", fn), tmp_path)
    a = data_loaders._open_dataset(str(tmp_path))
    assert os.path.exists(tmp_path / "toy.txe.npz") and not [f for f in os.listdir(tmp_path) if f.endswith(".tmp.npz")]
    monkeypatch.setattr(dataset, "read_terms", lambda *_: (_ for _ in ()).throw(AssertionError("raw files parsed again")))
    b = data_loaders._open_dataset(str(tmp_path))
    assert b.vocab == a.vocab and b.train_node_ids == a.train_node_ids and b.test_node_ids == a.test_node_ids
    assert torch.equal(b.g_full.ndata["x"], a.g_full.ndata["x"]) and np.array_equal(b.edges, a.edges)


def test_matcher_route_decision_and_deferred_vector_protocol_on_the_host():
    """model_zoo._Bilinear._route (ONE route per call, from the inputs alone) and the DeferredGraphVector protocol, without a GPU: the
    decisions that need no kernel -- RepeatedRows with / without repetition, the eval loop's expanded query, host tensors (never a run
    form), grad mode; can_fold / started / the refusal of a second differentiable use after the fold."""
    import numpy as np
    from taxoexpan_amd import model_zoo as mz, ops
    m = mz.LBM(6, 4)
    e1 = torch.randn(8, 6)
    table = torch.randn(3, 4)
    rep = ops.RepeatedRows.from_ids(table, np.repeat([0, 2], 4))                 # 2 distinct rows behind 8 pairs: repetition
    few = ops.RepeatedRows.from_ids(table, np.array([0, 1, 2, 0, 1, 2, 0, 1]))  # 8 runs of 1: none
    assert m._route(e1, rep)[0] == "runs"
    route, dense = m._route(e1, few)
    assert route == "pair" and torch.is_tensor(dense) and dense.shape == (8, 4)
    assert m._route(e1[:6], rep)[0] == "pair"                                    # row counts differ: densified
    q = torch.randn(4)
    with torch.no_grad():
        assert m._route(e1, q.expand(8, -1))[0] == "expand"                      # test_fast.py:122-123
    assert m._route(e1, q.expand(8, -1))[0] == "pair"                            # ... with gradients: the plain form
    assert m._route(e1, torch.randn(8, 4))[0] == "pair"                          # a host tensor never takes a device run form

    class _Node:                                                                 # stands in for a DeferredNodeOutput
        def __init__(self, ok):
            self._args = (type("C", (), dict(n_graphs=8, n_nodes=20, n_edges=30))(), None, torch.zeros(1), None, [])
            self._ok, self.calls = ok, []

        def _out_dim(self):
            return 6

        def _can_fold(self):
            return self._ok

        def _collapse(self, final, rpos, pw, fold_job=None):
            self.calls.append(final)
            c = type("Cfg", (), dict(link=ops.FoldLink()))()
            return ((torch.zeros(8, 32), torch.zeros(128, 32)) if final == "collapse_z" else torch.ones(8, 6)), c
    hv = mz.DeferredGraphVector(_Node(True), None, None)
    assert tuple(hv.shape) == (8, 6) and len(hv) == 8 and not hv.started() and hv.can_fold() and "pending" in repr(hv)
    assert m._route(hv, rep)[0] == "folded"
    with torch.no_grad():
        assert m._route(hv, rep)[0] == "runs"                                    # the fold is a training-time route
    t = hv.tensor()                                                              # any tensor consumer: the plain stack, once
    assert torch.is_tensor(t) and hv._src[0].calls == ["collapse"] and hv.started() and not hv.can_fold()
    assert hv.tensor() is t and m._route(hv, rep)[0] == "runs"
    assert not mz.DeferredGraphVector(_Node(False), None, None).can_fold()
    hv2 = mz.DeferredGraphVector(_Node(True), None, None)
    hv2._folded = True                                                           # (what match_folded leaves behind)
    with pytest.raises(RuntimeError, match="consumed FOLDED"):
        hv2.tensor()
    prev, ops._NO_MATCH_FOLD = ops._NO_MATCH_FOLD, True
    try:
        assert not mz.DeferredGraphVector(_Node(True), None, None).can_fold()
    finally:
        ops._NO_MATCH_FOLD = prev


def test_loss_tensor_backward_starts_from_a_unit_gradient_only_when_asked_plainly():
    """loss.LossTensor: `loss.backward()` hands autograd a cached constant 1 (on the device); anything else -- an explicit gradient,
    arithmetic on the loss, a host tensor -- is the ordinary torch path.  Here: the host-side mechanics (no GPU: the plain path)."""
    from taxoexpan_amd import loss as L
    x = torch.ones(3, requires_grad=True)
    l = (2 * x).sum().as_subclass(L.LossTensor)
    assert isinstance(l, L.LossTensor) and float(l.detach()) == 6.0
    l.backward()
    assert torch.equal(x.grad, torch.full((3,), 2.0))
    x.grad = None
    ((3 * x).sum().as_subclass(L.LossTensor) / 2).backward(torch.tensor(4.0))
    assert torch.equal(x.grad, torch.full((3,), 6.0))


def test_bench_compact_line_fits_the_drivers_tail_and_keeps_the_contract():
    """bench.compact_line on a real full record (profiles/r06_bench_full.json: what bench.py writes to bench_extra.json): the ONE stdout
    line stays far below the driver's 8 KB tail (round 5's 21 KB line parsed to null), carries the contract's keys with flat `roofline`
    and `cpu_baseline` objects, and -- with the N > 1 scalars present -- the multi-GPU keys of BASELINE configs[2] / [3]"""
    import importlib.util
    import json
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(repo, "profiles", "r06_bench_full.json")))
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 3000 < bench.COMPACT_LIMIT == 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "matrix_pipe", "value_fp32_mfma"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["vs_baseline"] is None
    assert set(line["config"]) >= {"workload", "egonets_per_step_per_gpu", "parallelism", "routes", "lr", "final_loss"}
    assert all(not isinstance(v, (dict, list)) for k in ("roofline", "cpu_baseline") for v in line[k].values())
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1 and len(line["cpu_baseline"]["sample"]) <= 240
    # N > 1: the flat keys ride along and the line still fits
    multi = dict(full, n_gpus=8, rccl_world=8, collective_backend="nccl", step_pgat2_dp_ms=3.4, step_pgat2_dp_edges_per_s=7.4e7,
                 candidates_scored_per_s_allgather=1.1e12, candidates_scored_per_s_fused_allreduce=2.2e12, allgather_gbs_per_rank=44.0,
                 allreduce_counts_queries_per_s=9.9e5)
    line8 = bench.compact_line(multi)
    for k in ("rccl_world", "collective_backend", "step_pgat2_dp_ms", "step_pgat2_dp_edges_per_s", "candidates_scored_per_s_allgather",
              "candidates_scored_per_s_fused_allreduce", "allgather_gbs_per_rank", "allreduce_counts_queries_per_s"):
        assert k in line8, k
    assert len(json.dumps(line8, separators=(",", ":"))) < 3500
