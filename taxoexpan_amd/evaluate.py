"""All-candidate evaluation of a trained model (test_fast.py:82-140, small-batch mode) on the MI355X path, end to end on device:
egonets of every candidate position built by `txe_egonet_*`, one encoder pass, fused scoring + ranking, and the reference's
per-query metric aggregation (`total_metrics[j] += metric(ranks of this query)`, divided by the number of queries)."""
import numpy as np
import torch

from .graph import device_egonet_batch
from .scoring import encode_candidates, rank_all_fused


def _per_query_means(values, pos_off):
    """mean over queries of the mean of `values` over the query's positives"""
    off = torch.as_tensor(pos_off).to(device=values.device, dtype=torch.int64)
    cnt = off[1:] - off[:-1]
    qid = torch.repeat_interleave(torch.arange(cnt.numel(), device=values.device), cnt)
    sums = torch.zeros(cnt.numel(), dtype=torch.float64, device=values.device).index_add_(0, qid, values.to(torch.float64))
    return float((sums / cnt.to(torch.float64)).mean().item())


def evaluate(model, dataset, device, larger_is_better=True, qblock=1024, seed=0):
    """dataset: taxoexpan_amd.dataset.MaskedGraphDataset in 'validation' or 'test' mode.  Returns (metrics dict, ranks int32
    [n_positives], pos_off [Q+1], queries list).  Queries whose true parents are not candidate positions are skipped, like the
    reference's rearrange() would fail on them."""
    device = torch.device(device)
    cand = sorted(dataset.all_positions)                                    # test_fast.py:93
    index = {a: i for i, a in enumerate(cand)}
    dtax = dataset.device_taxonomy(device)
    g = device_egonet_batch(dtax, np.asarray(cand, dtype=np.int64), expand_factor=dataset.expand_factor, seed=seed,
                            with_features="lazy")                           # x = features[_id]: the encoder projects the table once
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
    model.train(was_training)
    r = ranks.to(torch.float64)
    metrics = dict(macro_mr=_per_query_means(r, pos_off), hit_at_1=_per_query_means(ranks <= 1, pos_off),
                   hit_at_3=_per_query_means(ranks <= 3, pos_off), hit_at_5=_per_query_means(ranks <= 5, pos_off),
                   mrr_scaled_10=_per_query_means(1.0 / torch.ceil(r / 10.0), pos_off), n_queries=len(queries),
                   n_candidates=len(cand))
    return metrics, ranks, pos_off, queries
