"""Ranking metrics of model/cf/evaluator.go:75-160 (NDCG, Precision, Recall, HR, MAP, MRR) and the per-worker partial
sums of Evaluate (evaluator.go:35-72), on the host in float32 like the reference.

The reference gives every parallel.Parallel worker a partial sum per metric and a partial user count, adds the
partials and multiplies by 1 / count.  A rank of a multi-GPU job is one such worker (gorse_amd.dist.evaluate_sharded):
the rank lists come from the device (gorse_mf_rank = Rank + heap.TopKFilter), the metric arithmetic below is the
reference's, term by term, in float32."""
import numpy as np

F = np.float32


def _log2(x):
    return F(np.log2(F(x)))  # math32.Log2 on i + 2, i < topK: exact powers of two come out exact either way


def ndcg(target, rank_list):
    """evaluator.go:75-89"""
    target = set(int(t) for t in target)
    idcg = F(0)
    for i in range(min(len(target), len(rank_list))):
        idcg = F(idcg + F(1.0) / _log2(F(i) + F(2.0)))
    dcg = F(0)
    for i, item in enumerate(rank_list):
        if int(item) in target:
            dcg = F(dcg + F(1.0) / _log2(F(i) + F(2.0)))
    return F(dcg / idcg) if idcg != 0 else F(np.nan)  # Go: 0 / 0 = NaN for an empty rank list


def precision(target, rank_list):
    """evaluator.go:94-102"""
    target = set(int(t) for t in target)
    hit = F(sum(1 for item in rank_list if int(item) in target))
    return F(hit / F(len(rank_list))) if len(rank_list) else F(np.nan)


def recall(target, rank_list):
    """evaluator.go:108-116"""
    target = set(int(t) for t in target)
    hit = sum(1 for item in rank_list if int(item) in target)
    return F(F(hit) / F(len(target)))


def hr(target, rank_list):
    """evaluator.go:119-126"""
    target = set(int(t) for t in target)
    return F(1) if any(int(item) in target for item in rank_list) else F(0)


def mean_average_precision(target, rank_list):
    """evaluator.go:130-140"""
    target = set(int(t) for t in target)
    s, hit = F(0), 0
    for i, item in enumerate(rank_list):
        if int(item) in target:
            hit += 1
            s = F(s + F(hit) / F(i + 1))
    return F(s / F(len(target)))


def mrr(target, rank_list):
    """evaluator.go:153-160"""
    target = set(int(t) for t in target)
    for i, item in enumerate(rank_list):
        if int(item) in target:
            return F(F(1) / F(i + 1))
    return F(0)


METRICS = {"ndcg": ndcg, "precision": precision, "recall": recall, "hr": hr, "map": mean_average_precision, "mrr": mrr}


def candidates(test_ptr, test_idx, neg_ptr, neg_idx, users):
    """Per evaluated user: test positives followed by the sampled negatives (evaluator.go:52-55), as a CSR."""
    rows = [np.concatenate([test_idx[test_ptr[u]:test_ptr[u + 1]], neg_idx[neg_ptr[u]:neg_ptr[u + 1]]]) for u in users]
    ptr = np.zeros(len(rows) + 1, np.int64)
    if rows:
        np.cumsum([r.size for r in rows], out=ptr[1:])
    idx = np.concatenate(rows).astype(np.int32) if rows else np.zeros(0, np.int32)
    return ptr, idx


def partial_sums(rank, rank_len, users, test_ptr, test_idx, metrics=("ndcg", "precision", "recall")):
    """One worker's share of Evaluate: (float32 sum per metric over `users`, float32 user count), users in order.
    rank / rank_len: the device's rank lists for exactly these users (row r belongs to users[r])."""
    sums = np.zeros(len(metrics), F)
    count = F(0)
    for r, u in enumerate(users):
        target = test_idx[test_ptr[u]:test_ptr[u + 1]]
        if target.size == 0:
            continue
        rl = rank[r, :rank_len[r]]
        count = F(count + F(1))
        for m, name in enumerate(metrics):
            sums[m] = F(sums[m] + METRICS[name](target, rl))
    return sums, count
