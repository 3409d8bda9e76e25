"""GPU: the callers either side of the encoder -- infer.py's flow (new terms -> all nodes as candidates -> top-5 parents) and
test_fast.py's `-b` chunked evaluation -- against the same flows done literally on the host with the oracle."""
import os
import shutil

import numpy as np
import pytest
import torch

import txe_oracle as orc
from golden_util import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _toy(tmp_path, expand_factor=100):
    from taxoexpan_amd.dataset import MAGDataset, MaskedGraphDataset
    for fn in os.listdir(os.path.join(GOLDEN_DIR, "toy_taxo")):
        shutil.copy(os.path.join(GOLDEN_DIR, "toy_taxo", fn), tmp_path)
    return MaskedGraphDataset(MAGDataset("toy", str(tmp_path), raw=True), mode="test", sampling_mode=0, expand_factor=expand_factor,
                              normalize_embed=True)


def _model(match="LBM", prop="PGAT"):
    from taxoexpan_amd import TaxoExpan
    torch.manual_seed(11)
    return TaxoExpan(prop, "WMR", match, in_dim=8, hidden_dim=6, out_dim=5, pos_dim=3, num_layers=1, heads=[2, 1], feat_drop=0.1,
                     attn_drop=0.1, hidden_drop=0.1, out_drop=0.1).to(_dev())


def _oracle_hg(model, ds, anchors):
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    shapes, ids = [], []
    for a in anchors:
        nodes, k = ds._build_egonet(-1, a, 0)                      # infer.py:82 / test_fast.py:97: _get_subgraph(-1, anchor, 0)
        shapes.append((k, len(nodes) - k - 1))
        ids += nodes
    graph = orc.batch_egonets(shapes)
    hn = orc.pgat_forward(P, graph, ds.node_features[torch.tensor(ids)], [2, 1], 1, prefix="graph_propagate.")
    return orc.weighted_mean_readout(graph["graph_off"], hn, graph["pos"], P["readout.position_weights.weight"]), P


@pytest.mark.parametrize("match,loss,batch_size", [("LBM", "info_nce_loss", -1), ("BIM", "bce_loss", -1), ("LBM", "info_nce_loss", 17),
                                                   ("MLP", "bce_loss", 40)])
def test_infer_top5_parents_against_the_literal_loop(tmp_path, match, loss, batch_size):
    """infer.py:77-159 (small mode and `-b` chunks; descending for info_nce, ascending otherwise; any matcher)"""
    from taxoexpan_amd.evaluate import infer
    ds = _toy(tmp_path)
    model = _model(match)
    rs = np.random.RandomState(5)
    names = [f"new term {i}" for i in range(9)]
    vecs = rs.standard_normal((9, 8))
    taxon_file = tmp_path / "new.tsv"
    with open(taxon_file, "w") as f:
        for n, v in zip(names, vecs):
            f.write(n + "\t" + " ".join(repr(float(t)) for t in v) + "\n")
    save = tmp_path / "pred.tsv"
    out = infer(model, ds, str(taxon_file), _dev(), loss=loss, batch_size=batch_size, save=str(save))
    # host: the reference's loop with the oracle
    anchors = list(ds.graph.nodes())
    hg, P = _oracle_hg(model, ds, anchors)
    larger = loss.startswith("info_nce")
    want = []
    for v in vecs:
        q = torch.tensor(v, dtype=torch.float32).expand(len(anchors), -1)
        if match == "MLP":
            s = orc.mlp_match(hg, q, P["match.ffn.0.weight"], P["match.ffn.0.bias"], P["match.ffn.2.weight"], P["match.ffn.2.bias"])
        else:
            s = orc.bilinear_match(hg, q, P["match.W.weight"], match == "LBM")
        sc = s.squeeze(1).tolist()
        top = sorted(enumerate(sc), key=(lambda e: -e[1]) if larger else (lambda e: e[1]))[:5]
        want.append([ds.vocab[anchors[i]] for i, _ in top])
    assert [q for q, _ in out] == ["_".join(n.split(" ")) for n in names]              # infer.py:32
    # (scores of neighbouring ranks may differ by less than fp32 summation noise; then the ORDER may legitimately differ --
    #  require identical sets and identical order wherever the oracle's gaps are above 1e-5 relative)
    n_exact = 0
    for (qn, got), w in zip(out, want):
        assert len(got) == 5 and set(got) == set(w), (qn, got, w)
        n_exact += got == w
    assert n_exact >= len(want) - 1
    lines = open(save).read().splitlines()
    assert lines[0] == "Query\tPredicted parents" and lines[1] == f"{out[0][0]}\t{', '.join(out[0][1])}" and len(lines) == 10


@pytest.mark.parametrize("match,larger", [("LBM", True), ("BIM", False)])
def test_case_study_table_against_the_literal_loop(tmp_path, match, larger):
    """test_fast.py:112-147 (`-c`): per test query its name, true parents, top-5 predicted parents and every metric on that query
    alone -- evaluate(case=...) against the reference's loop done literally on the host with the oracle (scores, metric.py ranks,
    Python's stable sort), and the TSV it writes"""
    from taxoexpan_amd.evaluate import CASE_METRICS, evaluate
    ds = _toy(tmp_path)
    model = _model(match)
    path = tmp_path / "case.tsv"
    metrics, ranks, pos_off, queries = evaluate(model, ds, _dev(), larger_is_better=larger, case=str(path))
    rows = [ln.split("\t") for ln in open(path).read().splitlines()]
    assert rows[0] == ["Test node index", "True parents", "Predicted parents"] + list(CASE_METRICS)
    assert len(rows) == 1 + len(queries) and len(queries) >= 5
    cand = sorted(ds.all_positions)                                               # test_fast.py:93
    hg, P = _oracle_hg(model, ds, cand)
    index = {a: i for i, a in enumerate(cand)}
    n_exact = 0
    for row, q in zip(rows[1:], queries):
        qv = ds.node_features[q].expand(len(cand), -1)                            # test_fast.py:121-123
        sc = orc.bilinear_match(hg, qv, P["match.W.weight"], match == "LBM").squeeze(1).tolist()
        top = sorted(enumerate(sc), key=(lambda e: -e[1]) if larger else (lambda e: e[1]))[:5]
        pos = [index[a] for a in ds.node2parents[q] if a in index]
        r = np.asarray(orc.ranks_of_positives(sc, pos, larger), dtype=np.float64)  # metric.py:7-31
        want = [ds.vocab[q], ", ".join(ds.vocab[a] for a in ds.node2parents[q]), ", ".join(ds.vocab[cand[i]] for i, _ in top),
                str(float(r.mean())), str(float(r.mean())), str(float(np.sum(r <= 1) / len(r))), str(float(np.sum(r <= 3) / len(r))),
                str(float(np.sum(r <= 5) / len(r))), str(float((1.0 / np.ceil(r / 10)).mean()))]
        assert row[:2] == want[:2] and row[3:] == want[3:], (row, want)
        assert set(row[2].split(", ")) == set(want[2].split(", "))
        n_exact += row[2] == want[2]
    assert n_exact >= len(queries) - 1                # (neighbouring scores inside fp32 summation noise may swap)
    rows2 = []
    evaluate(model, ds, _dev(), larger_is_better=larger, case=rows2)
    assert rows2 == rows


def test_newterm_magnitudes_scores_and_top5_on_device():
    """infer.py:23-38,96-106 with query vectors of data/mag_cs_new637.txt's magnitude (rows divided by their SUM: entries up to ~270):
    the factored scoring GEMM with the fused exp against the reference's LBM / BIM scores (tests/golden/newterms.npz) -- infs where exp
    overflows, zeros where it underflows -- and scoring.topk_parents on the DEVICE scores against the reference's sorted() top-5, both
    directions: equal infs / zeros / duplicated candidates in candidate order"""
    import types
    from taxoexpan_amd import ops
    from taxoexpan_amd.scoring import topk_parents, topk_parents_fused
    z = dict(np.load(os.path.join(GOLDEN_DIR, "newterms.npz")))
    import golden_cases as gc
    hg, _raw, W = gc.make_newterm_inputs()
    dev = _dev()
    hg_d, W_d, q_d = torch.from_numpy(hg).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(z["nf32"]).to(dev)
    ids = torch.arange(hg.shape[0], device=dev)
    U = ops.bilinear_project(hg_d, W_d)
    for kind, ex in (("lbm", True), ("bim", False)):
        S = ops.score_block(q_d, U, ex)
        ref = z[f"S_{kind}"]
        got = S.cpu().numpy()
        # overflow / underflow happen at the same entries unless the exponent sits within 1e-4 relative of the fp32 thresholds
        edge = np.zeros_like(ref, dtype=bool)
        if ex:
            with np.errstate(over="ignore"):
                s64 = np.einsum("gl,lr,qr->qg", hg.astype(np.float64), W[0].astype(np.float64), z["nf32"].astype(np.float64))
            edge = (np.abs(s64 - 88.7228) < 2e-2) | (np.abs(s64 + 103.28) < 2e-1) | (np.abs(s64 + 87.3365) < 2e-2)
        assert np.array_equal(np.isinf(got) | edge, np.isinf(ref) | edge)
        fin = np.isfinite(ref) & np.isfinite(got) & (ref != 0) & ~edge & (np.abs(ref) > 1e-30)
        # BIM: bilinear values up to ~140 built from cancelling terms -- the noise floor is absolute, 2e-6 of the largest value (the fp32
        # summation order differs); LBM: exp turns that absolute noise of the exponent into a relative one
        if ex:
            np.testing.assert_allclose(got[fin], ref[fin], rtol=3e-4)
        else:
            np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-4, atol=2e-6 * float(np.abs(ref[fin]).max()))
        for larger, key in ((True, "desc"), (False, "asc")):
            top = topk_parents(S, ids, 5, larger).cpu().numpy()
            want = z[f"top5_{key}_{kind}"]
            n_same = 0
            for qi in range(top.shape[0]):
                mine = sorted(enumerate(got[qi].tolist()), key=(lambda e: -e[1]) if larger else (lambda e: e[1]))[:5]
                assert top[qi].tolist() == [e[0] for e in mine]                              # the device scores' own stable sort
                n_same += top[qi].tolist() == want[qi].tolist()
            assert n_same >= top.shape[0] - 1, (kind, key, n_same)                            # = the reference's, up to one noise swap
        inf_rows = np.nonzero(np.isinf(ref).sum(1) >= 5)[0]
        if ex:
            assert len(inf_rows) >= 2
            t = topk_parents(S, ids, 5, True).cpu().numpy()
            assert np.array_equal(t[inf_rows], z["top5_desc_lbm"][inf_rows])               # >= 5 equal infs: exactly candidate order
        # the fused score + select kernels (no score matrix) make the SAME selection as the composite on the materialised scores of the
        # same GEMM kernel -- bit-identical values: overflowed infs / underflowed zeros / duplicated candidates in candidate order
        mod = types.SimpleNamespace(W=types.SimpleNamespace(weight=W_d), apply_exp=ex)
        for larger in (True, False):
            for k in (1, 5, 8):
                fused = topk_parents_fused(mod, hg_d, q_d, ids, k, larger)
                assert torch.equal(fused, topk_parents(S, ids, k, larger)), (kind, larger, k)
            assert torch.equal(topk_parents_fused(mod, hg_d, q_d, ids, 5, larger, block=7), topk_parents(S, ids, 5, larger))   # ragged query blocks


@pytest.mark.parametrize("batch_size", [13, 64])
def test_chunked_evaluation_equals_single_batch(tmp_path, batch_size):
    """test_fast.py:149-218 (`-b`): the candidates encoded in chunks give the ranks and metrics of the one-batch run; expand_factor
    below the largest child count, so sampled siblings must be identical too (index_base)"""
    from taxoexpan_amd.evaluate import evaluate
    ds = _toy(tmp_path, expand_factor=3)
    model = _model("LBM")
    m1, r1, off1, q1 = evaluate(model, ds, _dev(), seed=3)
    m2, r2, off2, q2 = evaluate(model, ds, _dev(), seed=3, batch_size=batch_size)
    assert torch.equal(r1.cpu(), r2.cpu()) and list(off1) == list(off2) and q1 == q2
    assert m1 == m2


def test_mag_full_encode_in_30000_chunks_equals_single_batch():
    """BASELINE configs[2]'s `batch_size=30000` on the MAG-Full shape: 356 k candidate egonets encoded in 12 chunks of 30,000
    (test_fast.py:149-179) against the single 1.1 M-node batch -- same table projection, same sampled siblings"""
    from taxoexpan_amd import TaxoExpan, graph as G, synthetic as syn
    from taxoexpan_amd.evaluate import candidate_graphs
    from taxoexpan_amd.scoring import encode_candidates
    dev = _dev()
    tax = syn.make_named_taxonomy("mag_full", seed=47)
    cand, _val, _test = syn.split_candidates(tax)
    torch.manual_seed(47)
    model = TaxoExpan("PGAT", "WMR", "LBM", in_dim=250, hidden_dim=500, out_dim=500, pos_dim=50, num_layers=1, heads=[4, 1], feat_drop=0.1,
                      attn_drop=0.1, hidden_drop=0.1, out_drop=0.1).to(dev).eval()
    dtax = G.DeviceTaxonomy(tax.par_ptr, tax.par_idx, tax.chd_ptr, tax.chd_idx, tax.features, dev)
    chunks = candidate_graphs(dtax, cand, 50, 7, batch_size=30000)
    assert isinstance(chunks, list) and len(chunks) == -(-len(cand) // 30000) == 12
    hg_c = encode_candidates(model, chunks)
    del chunks
    one = candidate_graphs(dtax, cand, 50, 7, batch_size=-1)
    assert int(one.number_of_nodes()) > 1_000_000
    hg_1 = encode_candidates(model, one)
    assert hg_c.shape == hg_1.shape == (len(cand), 500)
    torch.cuda.synchronize()
    d = (hg_c - hg_1).abs().max().item()
    assert d <= 1e-5 * hg_1.abs().max().item() + 1e-6, d


def test_bench_two_ranks_sharded_inference_on_one_gpu():
    """bench.py's N > 1 path end to end on a single-GPU box: two ranks (both on cuda:0) over gloo -- data-parallel training step
    with the overlapped gradient all-reduce, then extra_metrics_sharded: candidate-sharded MAG-Full inference with the pipelined
    all-gather of score blocks and the all-reduce-of-counts ranking.  (RCCL itself needs one GPU per rank: world size 1 is covered
    in test_gpu_parity.py, the 8-GPU run is the driver's.)"""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TXE_BENCH_BACKEND="gloo", TXE_BENCH_SHARDED_QUERIES="2048", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    # the compact line says which collective transport ran on how many ranks, and carries BASELINE's multi-GPU configs flat:
    # configs[3] (2-layer MAG-Full step, data-parallel) and configs[2]'s sharded scoring with its all-gather / all-reduce rates
    assert d["rccl_world"] == 2 and d["collective_backend"] == "gloo"
    for k in ("step_pgat2_dp_ms", "step_pgat2_dp_edges_per_s", "candidates_scored_per_s_allgather", "candidates_scored_per_s_fused_allreduce",
              "allgather_gbs_per_rank", "allreduce_counts_queries_per_s"):
        assert d[k] > 0, k
    with open(os.path.join(repo, "bench_extra.json")) as f:          # (the full record of the same run; stderr of two ranks may interleave)
        full = json.load(f)
    assert full["value"] == d["value"] and full["n_gpus"] == 2
    ex = full["extra"]
    assert "error" not in ex and "error" not in ex["step_pgat2_dp"], ex
    assert "dp2" in ex["step_pgat2_dp"]["workload"] and ex["step_pgat2_dp"]["gradient_bytes_per_rank_per_step"] > 0
    assert "error" not in ex, ex
    assert ex["infer_queries"] == 2048 and ex["candidates_per_rank"] * 2 >= ex["infer_candidates"]
    for k in ("candidates_scored_per_s_local", "candidates_scored_per_s_allgather", "candidates_scored_per_s_fused_allreduce",
              "infer_top5_queries_per_s_sharded"):
        assert ex[k] > 0, k
    assert ex["infer_top5_shape"] == [2048, 5]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_scoring_equals_unsharded_bit_for_bit_on_the_hip_kernels(world):
    """candidate-sharded scoring with the HIP kernels at world size 2 and 4 (all ranks on cuda:0, gloo): the pipelined all-gather of
    score blocks read in place and densified, and the all-reduce-of-counts ranking, against the unsharded loop of the same process --
    torch.equal (tests/dist_gpu_worker.py).  The CPU tests prove the collective logic with injected local functions; this one runs
    the kernels under it."""
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(repo, "tests", "dist_gpu_worker.py")],
                         cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert all(out.stdout.count(f"OK {r}") == 1 for r in range(world)), out.stdout[-500:]      # (the ranks' lines may interleave)


@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_training_step_equals_single_process_on_the_hip_kernels(world):
    """trainer/trainer.py:52-56 + model/loss.py:52-57 under data parallelism, with the real TaxoExpan / GATStackFunction announcing its
    buckets into overlapped_gradient_allreduce(model=...): world 2 and 4 over gloo on one GPU, each rank a query shard of ONE batch;
    every parameter gradient equals (a) the sum of the shards' gradients computed without any collective, to 2e-6 (the collectives add
    nothing but the sum), and that sum equals (b) the single-process step on the whole batch; two- and three-layer PGAT (one / two
    planned buckets); a rank with an EMPTY shard issues the planned collectives with zeros and nothing hangs
    (tests/dist_gpu_worker.py dp)."""
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(repo, "tests", "dist_gpu_worker.py"), "dp"],
                         cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, "\n".join(ln for ln in out.stderr.splitlines() if ln.startswith("[rank"))[-3000:]
    assert all(out.stdout.count(f"OK {r}") == 1 for r in range(world)), out.stdout[-500:]


@pytest.mark.parametrize("mode", ["score", "dp"])
def test_rccl_world2_sharded_scoring_and_data_parallel_step(mode):
    """the two multi-process tests above over RCCL (backend "nccl"), one GPU per rank, world size 2: the transport the 8-GPU runs use --
    async collectives on RCCL's own streams, `work.wait()` as a stream dependency.  Needs two GPUs: skipped on the one-GPU boxes the
    round's tests run on (there the gloo variants cover the logic over the same kernels)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL with one GPU per rank)")
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TXE_TEST_BACKEND="nccl")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(repo, "tests", "dist_gpu_worker.py")] + (["dp"] if mode == "dp" else []),
                         cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert all(out.stdout.count(f"OK {r}") == 1 for r in range(2)), out.stdout[-500:]


@pytest.mark.parametrize("G,Q,k", [(24736, 700, 5), (131, 40, 8), (128, 3, 5), (5, 9, 8), (1000, 260, 3)])
def test_fused_top_k_equals_the_composite_on_materialised_scores(G, Q, k):
    """txe_score_topk_block + txe_topk_merge (infer.py:96-106, test_fast.py:121-131 without the score matrix) against
    scoring.topk_parents -- Python's stable sort -- on the scores txe_score_block materialises with the same kernel: the MAG-CS
    candidate count (194 column tiles, a ragged last one), fewer candidates than one tile / than k, duplicated candidate rows (exact
    ties across tiles: candidate order), NaN and +-inf scores, both directions, a candidate shard offset, and the merge of per-shard
    lists against the unsharded selection"""
    from taxoexpan_amd import model_zoo as mz, ops
    from taxoexpan_amd.scoring import topk_parents, topk_parents_fused
    dev = _dev()
    gen = torch.Generator().manual_seed(G + Q)
    l, r = 500, 250
    hg = torch.randn(G, l, generator=gen) * 0.3
    hg[G // 2:] = hg[:G - G // 2].clone()                          # every candidate row twice: exact ties in different tiles
    if G > 200:
        hg[7] = float("nan")                                       # a NaN score ranks last
        hg[11] *= 1e4                                              # +-inf after exp / huge values
    queries = torch.nn.functional.normalize(torch.randn(Q, r, generator=gen), dim=1)
    if G == 1000:
        hg[:] = hg[0]                                              # EVERY score of a query equal: the k lowest candidate positions, whatever
                                                                   # the tiles' floors did in between
    ids = torch.arange(G, device=dev) * 3 + 1
    for kind in ("LBM", "BIM"):
        torch.manual_seed(5)
        match = getattr(mz, kind)(l, r).to(dev)
        with torch.no_grad():
            U = ops.bilinear_project(hg.to(dev), match.W.weight)
            S = ops.score_block(queries.to(dev), U, match.apply_exp)
            for larger in (True, False):
                want = topk_parents(S, ids, k, larger)
                got = topk_parents_fused(match, hg.to(dev), queries.to(dev), ids, k, larger)
                assert got.shape == want.shape == (Q, min(k, G)) and torch.equal(got, want), (kind, larger)
                if G == 1000:
                    assert torch.equal(got, ids[:k].expand(Q, k)), (kind, larger)
                if G >= 16:                                        # two "shards" merged = the unsharded selection
                    h = G // 3
                    parts = []
                    for lo, hi in ((0, h), (h, G)):
                        Us = ops.bilinear_project(hg[lo:hi].to(dev), match.W.weight)
                        parts.append(ops.score_topk_block(queries.to(dev), Us, match.apply_exp, min(k, hi - lo), larger, idx_base=lo))
                    kk = min(k, h)
                    cat_i = torch.cat([p[0][:, :kk] for p in parts], 1)
                    cat_k = torch.cat([p[1][:, :kk] for p in parts], 1)
                    idx, _key = ops.topk_merge(cat_k, cat_i, kk)
                    assert torch.equal(ids[idx.long()], topk_parents(S, ids, kk, larger)), (kind, larger, "shards")
