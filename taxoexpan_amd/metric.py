"""model/metric.py of the reference on the flat rank layout the device ranking produces.

The reference passes `all_ranks` = a list (one entry per query) of lists of ranks; `ops.rank_block` / `scoring.rank_all_fused`
return one int32 tensor `ranks [n_positives]` plus the offsets `pos_off [Q+1]` of each query's positives.  Every function below
takes that pair and computes, on whatever device the ranks live, what the same-named reference function computes
(metric.py:62-96): macro_mr, micro_mr, hit_at_1/3/5, mrr_scaled_10, combined_metrics.  Results are Python floats.
`as_rank_lists` converts back to the reference's nested-list layout.
"""
import torch


def _f(ranks):
    return ranks.to(torch.float64)


def _counts(pos_off, device):
    off = torch.as_tensor(pos_off).to(device=device, dtype=torch.int64)
    return off, off[1:] - off[:-1]


def as_rank_lists(ranks, pos_off):
    """the reference's `all_ranks` (list of per-query rank lists), e.g. to call the reference's own metric functions"""
    r = ranks.cpu().tolist()
    off = torch.as_tensor(pos_off).cpu().tolist()
    return [r[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def macro_mr(ranks, pos_off):
    """metric.py:62-64: mean over queries of the query's mean rank"""
    off, cnt = _counts(pos_off, ranks.device)
    qid = torch.repeat_interleave(torch.arange(cnt.numel(), device=ranks.device), cnt)
    sums = torch.zeros(cnt.numel(), dtype=torch.float64, device=ranks.device).index_add_(0, qid, _f(ranks))
    return float((sums / cnt.to(torch.float64)).mean().item())


def micro_mr(ranks, pos_off=None):
    """metric.py:66-68: mean over all positives"""
    return float(_f(ranks).mean().item())


def _hit(ranks, k):
    return float((ranks <= k).to(torch.float64).mean().item())


def hit_at_1(ranks, pos_off=None):
    return _hit(ranks, 1)


def hit_at_3(ranks, pos_off=None):
    return _hit(ranks, 3)


def hit_at_5(ranks, pos_off=None):
    return _hit(ranks, 5)


def mrr_scaled_10(ranks, pos_off=None):
    """metric.py:85-90: mean of 1 / ceil(rank / 10)"""
    return float((1.0 / torch.ceil(_f(ranks) / 10.0)).mean().item())


def combined_metrics(ranks, pos_off):
    """metric.py:92-96 (early-stopping score)"""
    return (macro_mr(ranks, pos_off) * (1.0 / max(mrr_scaled_10(ranks), 0.0001)) * (1.0 / max(hit_at_3(ranks), 0.0001)) *
            (1.0 / max(hit_at_1(ranks), 0.0001)))
