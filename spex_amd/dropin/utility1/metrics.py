"""Ranking metrics with the reference's definitions (LightGCN_SPEX/code/utility1/metrics.py), vectorised.

`r` is the 0/1 relevance list of the returned top-K_max items, best first.
  recall_at_k  metrics.py:74-80   hits in the first k / number of positives (one held-out positive => HR@k)
  dcg_at_k     metrics.py:43-58   sum r_i / log2(i + 2)                (method 1)
  ndcg_at_k    metrics.py:61-71   DCG / DCG of the *returned list* sorted descending
"""
import numpy as np


def _as_float(r):
    return np.asarray(r, dtype=np.float64)


def dcg_at_k(r, k, method=1):
    r = _as_float(r)[:k]
    if not r.size:
        return 0.0
    if method == 1:
        return float(np.sum(r / np.log2(np.arange(2, r.size + 2))))
    if method == 0:
        return float(r[0] + np.sum(r[1:] / np.log2(np.arange(2, r.size + 1))))
    raise ValueError("method must be 0 or 1.")


def ndcg_at_k(r, k, method=1):
    ideal = dcg_at_k(sorted(r, reverse=True), k, method)
    return dcg_at_k(r, k, method) / ideal if ideal else 0.0


def recall_at_k(r, k, all_pos_num):
    return float(np.sum(_as_float(r)[:k]) / all_pos_num) if all_pos_num else 0.0


def precision_at_k(r, k):
    assert k >= 1
    return float(np.mean(np.asarray(r)[:k]))


def hit_at_k(r, k):
    return 1.0 if np.sum(np.asarray(r)[:k]) > 0 else 0.0


def F1(pre, rec):
    return (2.0 * pre * rec) / (pre + rec) if pre + rec > 0 else 0.0


def rank_metrics_batch(rel, ks, n_pos):
    """recall@k and ndcg@k for many users at once.  rel: [n, K_max] 0/1, best first; n_pos: [n] positives per user.
    Same numbers as calling the scalar functions per user."""
    rel = np.asarray(rel, np.float64)
    n, kmax = rel.shape
    disc = 1.0 / np.log2(np.arange(2, kmax + 2))
    ideal_rel = -np.sort(-rel, axis=1)
    recall, ndcg = np.zeros((n, len(ks))), np.zeros((n, len(ks)))
    for j, k in enumerate(ks):
        k = min(k, kmax)
        hits = rel[:, :k].sum(1)
        recall[:, j] = np.divide(hits, n_pos, out=np.zeros(n), where=np.asarray(n_pos) > 0)
        dcg = (rel[:, :k] * disc[:k]).sum(1)
        ideal = (ideal_rel[:, :k] * disc[:k]).sum(1)
        ndcg[:, j] = np.divide(dcg, ideal, out=np.zeros(n), where=ideal > 0)
    return recall, ndcg
