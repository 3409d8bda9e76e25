"""All-candidate evaluation (test_fast.py:82-225) and inference on new terms (infer.py:77-159) on the MI355X path, end to end on
device: egonets of every candidate position built by `txe_egonet_*`, the encoder in one batch (`-b -1`) or in chunks of `-b`
egonets, then fused scoring + ranking with the reference's per-query metric aggregation (`evaluate`), or the top-5 parents of every
new term (`infer`)."""
import numpy as np
import torch

from .graph import device_egonet_batch
from .scoring import encode_candidates, rank_all_fused, topk_parents, topk_parents_fused


def candidate_graphs(dtax, anchors, expand_factor, seed, batch_size=-1):
    """the candidate egonets `_get_subgraph(-1, anchor, 0)` of test_fast.py:93-97 / infer.py:80-82 as one device-built batch
    (batch_size == -1: the scripts' small mode) or as chunks of batch_size egonets (`-b`, test_fast.py:149-179 / infer.py:108-139)"""
    anchors = np.asarray(anchors, dtype=np.int64)
    if batch_size is None or batch_size <= 0 or batch_size >= len(anchors):
        return device_egonet_batch(dtax, anchors, expand_factor=expand_factor, seed=seed, with_features="lazy")
    return [device_egonet_batch(dtax, anchors[i:i + batch_size], expand_factor=expand_factor, seed=seed, with_features="lazy", index_base=i)
            for i in range(0, len(anchors), batch_size)]


def _per_query_means(values, pos_off):
    """mean over queries of the mean of `values` over the query's positives"""
    off = torch.as_tensor(pos_off).to(device=values.device, dtype=torch.int64)
    cnt = off[1:] - off[:-1]
    qid = torch.repeat_interleave(torch.arange(cnt.numel(), device=values.device), cnt)
    sums = torch.zeros(cnt.numel(), dtype=torch.float64, device=values.device).index_add_(0, qid, values.to(torch.float64))
    return float((sums / cnt.to(torch.float64)).mean().item())


CASE_METRICS = ("macro_mr", "micro_mr", "hit_at_1", "hit_at_3", "hit_at_5", "mrr_scaled_10")     # config.mag.json "metrics"


def _score_blocks(model, hg, qf, qblock):
    """score blocks [<= qblock queries, G candidates] of the per-query loop test_fast.py:121-123 / infer.py:96-98: one factored GEMM per
    block for BIM / LBM, the literal expand loop for any other matcher"""
    from . import ops
    U = None
    for q0 in range(0, qf.shape[0], qblock):
        if hasattr(model.match, "W") and hasattr(model.match, "apply_exp"):
            U = ops.bilinear_project(hg, model.match.W.weight) if U is None else U
            yield q0, ops.score_block(qf[q0:q0 + qblock], U, model.match.apply_exp)
        else:
            yield q0, torch.stack([model.match(hg, q.expand(hg.shape[0], -1)).reshape(-1) for q in qf[q0:q0 + qblock]])


def _best_parents(model, hg, qf, cand_ids, topk, larger_is_better, qblock):
    """the `topk` best candidates of every query, best first (infer.py:100-106 / test_fast.py:125-131): BIM / LBM with topk <= 8 through
    the fused score + select kernels (no score matrix), anything else by materialising score blocks"""
    if hasattr(model.match, "W") and hasattr(model.match, "apply_exp") and 1 <= topk <= 8 and hg.shape[0] > 0 and qf.shape[0] > 0:
        return topk_parents_fused(model.match, hg, qf, cand_ids, topk, larger_is_better, block=qblock)
    top = [topk_parents(S, cand_ids, topk, larger_is_better) for _q0, S in _score_blocks(model, hg, qf, qblock or 1024)]
    return torch.cat(top) if top else cand_ids.new_zeros((0, 0))


def _case_rows(dataset, queries, pos_off, ranks, top, metric_names):
    """the case-study table of test_fast.py:112-147: per test query its name, true parents, predicted top-5 parents and every
    metric evaluated on that query's ranks alone (`metric([ranks])`, model/metric.py:62-90), as strings"""
    r = ranks.cpu().to(torch.float64).numpy()
    per_query = {
        "macro_mr": lambda x: float(x.mean()), "micro_mr": lambda x: float(x.mean()),
        "hit_at_1": lambda x: float(1.0 * np.sum(x <= 1) / len(x)), "hit_at_3": lambda x: float(1.0 * np.sum(x <= 3) / len(x)),
        "hit_at_5": lambda x: float(1.0 * np.sum(x <= 5) / len(x)), "mrr_scaled_10": lambda x: float((1.0 / np.ceil(x / 10)).mean()),
    }
    vocab = dataset.vocab
    rows = [["Test node index", "True parents", "Predicted parents"] + list(metric_names)]
    for i, q in enumerate(queries):
        x = r[pos_off[i]:pos_off[i + 1]]
        rows.append([vocab[q], ", ".join(vocab[p] for p in dataset.node2parents[q]), ", ".join(vocab[p] for p in top[i])] +
                    [str(per_query[m](x)) for m in metric_names])
    return rows


def evaluate(model, dataset, device, larger_is_better=True, qblock=None, seed=0, batch_size=-1, case=None, metric_names=CASE_METRICS,
             topk=5):
    """dataset: taxoexpan_amd.dataset.MaskedGraphDataset in 'validation' or 'test' mode.  Returns (metrics dict, ranks int32
    [n_positives], pos_off [Q+1], queries list).  Queries whose true parents are not candidate positions are skipped, like the
    reference's rearrange() would fail on them.
    case: test_fast.py's `-c` -- a path (the TSV of :142-147 is written) or a list (the rows are appended): per query its name, true
    parents, the `topk` predicted parents (best first: descending score when larger_is_better, i.e. the info_nce losses, ascending
    otherwise; ties in candidate order like Python's stable sort) and the metrics of `metric_names` on that query alone."""
    device = torch.device(device)
    cand = sorted(dataset.all_positions)                                    # test_fast.py:93
    index = {a: i for i, a in enumerate(cand)}
    dtax = dataset.device_taxonomy(device)
    g = candidate_graphs(dtax, cand, dataset.expand_factor, seed, batch_size)    # x = features[_id]: the table is projected once
    was_training = model.training
    model.eval()
    hg = encode_candidates(model, g)                                        # test_fast.py:99-108
    queries, pos_lists = [], []
    for q in dataset.node_list:
        p = [index[a] for a in dataset.node2parents[q] if a in index]
        if p:
            queries.append(q)
            pos_lists.append(p)
    pos_off = np.concatenate([[0], np.cumsum([len(p) for p in pos_lists])]).astype(np.int64)
    pos_idx = np.concatenate(pos_lists).astype(np.int64) if pos_lists else np.zeros(0, dtype=np.int64)
    qf = dataset.node_features[torch.as_tensor(queries, dtype=torch.long)].to(device)
    with torch.no_grad():
        ranks = rank_all_fused(model.match, hg, qf, pos_off, pos_idx, block=qblock, larger_is_better=larger_is_better)
        if case is not None:                                               # test_fast.py:112-147
            cand_ids = torch.as_tensor(np.asarray(cand, dtype=np.int64), device=device)
            top = _best_parents(model, hg, qf, cand_ids, topk, larger_is_better, qblock).cpu().tolist()
            rows = _case_rows(dataset, queries, pos_off, ranks, top, metric_names)
            if isinstance(case, list):
                case.extend(rows)
            else:
                with open(case, "w") as fout:
                    for row in rows:
                        fout.write("\t".join(row))
                        fout.write("\n")
    model.train(was_training)
    r = ranks.to(torch.float64)
    metrics = dict(macro_mr=_per_query_means(r, pos_off), hit_at_1=_per_query_means(ranks <= 1, pos_off),
                   hit_at_3=_per_query_means(ranks <= 3, pos_off), hit_at_5=_per_query_means(ranks <= 5, pos_off),
                   mrr_scaled_10=_per_query_means(1.0 / torch.ceil(r / 10.0), pos_off), n_queries=len(queries),
                   n_candidates=len(cand))
    return metrics, ranks, pos_off, queries


def infer(model, dataset, new_taxons, device, loss="info_nce_loss", batch_size=-1, save=None, topk=5, normalize=False, qblock=1024,
          seed=0):
    """infer.py:77-159: the `topk` best parents of every NEW term.  dataset: MaskedGraphDataset in 'test' mode (infer.py:43-57);
    new_taxons: path of the `<name>\t<v0 v1 ...>` file (infer.py:23-38) or an already loaded (vocab, array) pair.  Candidates are ALL
    nodes of the dataset's graph (infer.py:80-82 iterates `test_dataset.graph.nodes()`, not `all_positions`), in node order; best =
    descending score for the info_nce losses, ascending otherwise (infer.py:100-106), ties in candidate order like Python's stable
    sort.  Returns [(query, [parent vocab entries])]; `save` writes infer.py's TSV (header `Query\tPredicted parents`)."""
    from .dataset import load_new_taxons
    device = torch.device(device)
    vocab, nf = load_new_taxons(new_taxons, normalize) if isinstance(new_taxons, (str, bytes)) or hasattr(new_taxons, "__fspath__") else new_taxons
    anchors = np.asarray(list(dataset.graph.nodes), dtype=np.int64)
    dtax = dataset.device_taxonomy(device)
    g = candidate_graphs(dtax, anchors, dataset.expand_factor, seed, batch_size)
    was_training = model.training
    model.eval()
    hg = encode_candidates(model, g)
    larger = str(loss).startswith("info_nce")
    qf = torch.as_tensor(np.asarray(nf), dtype=torch.float32).to(device)
    cand_ids = torch.as_tensor(anchors, device=device)
    with torch.no_grad():
        picks = _best_parents(model, hg, qf, cand_ids, topk, larger, qblock).cpu().tolist()
    model.train(was_training)
    out = [(q, [dataset.vocab[i] for i in row]) for q, row in zip(vocab, picks)]
    if save is not None:
        with open(save, "w") as fout:
            fout.write("Query\tPredicted parents\n")
            for q, parents in out:
                fout.write(f"{q}\t{', '.join(parents)}\n")
    return out
