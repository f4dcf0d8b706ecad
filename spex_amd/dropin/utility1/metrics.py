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


def ranked_relevance(items, scores, positives, k_max):
    """The reference's `ranklist_by_heapq` (utility1/batch_test.py:80-90, NGCF utility/batch_test.py:34-49): candidates
    go into a dict item -> score (a repeated item keeps its first position and its last score), heapq.nlargest keeps
    the k_max best with ties in insertion order (== a stable descending sort), and the result is the 0/1 membership of
    each returned item in `positives`."""
    rating = {}
    for it, sc in zip(items, scores):
        rating[it] = sc
    keys = list(rating)
    vals = np.asarray([rating[k] for k in keys], np.float64)
    order = np.argsort(-vals, kind="stable")[:k_max]
    pos = set(positives)
    return [1 if keys[j] in pos else 0 for j in order]


def accumulate_rank_metrics(cand, scores, positives, ks):
    """Sum over users of recall@k / ndcg@k divided by the number of users, accumulated user by user in float64 — the
    order and dtype of utility1/batch_test.py:19-24 and NGCF utility/batch_test.py:166-168.
    cand: per user the candidate item ids (negatives first, then the positives); scores: flat array of their scores in
    the same order; positives: per user the list of held-out items.  Users whose candidates are distinct, equally many
    and have one positive take a vectorised path; everyone else the dict semantics above."""
    n = len(cand)
    result = {"recall": np.zeros(len(ks)), "ndcg": np.zeros(len(ks))}
    if n == 0:
        return result
    kmax = max(ks)
    lens = np.array([len(c) for c in cand])
    start = np.concatenate([[0], np.cumsum(lens)])
    scores = np.asarray(scores, np.float64)
    rel = np.zeros((n, kmax))
    n_pos = np.ones(n)
    fast = np.zeros(n, bool)
    if lens.min() == lens.max() and lens[0] > 0:
        L = int(lens[0])
        it = np.concatenate([np.asarray(c, np.int64) for c in cand]).reshape(n, L)
        sc = scores.reshape(n, L)
        srt = np.sort(it, axis=1)
        fast = ~(srt[:, 1:] == srt[:, :-1]).any(1) if L > 1 else np.ones(n, bool)
        fast &= np.array([len(p) == 1 for p in positives])
        order = np.argsort(-sc, axis=1, kind="stable")[:, :kmax]
        top_items = np.take_along_axis(it, order, 1)
        pos_item = np.array([p[0] if len(p) else -1 for p in positives])
        rel[:, :top_items.shape[1]] = top_items == pos_item[:, None]
    rec_f, ndcg_f = rank_metrics_batch(rel, ks, n_pos)
    for j in range(n):
        if fast[j]:
            rec, nd = rec_f[j], ndcg_f[j]
        else:
            r = ranked_relevance(cand[j], scores[start[j]:start[j + 1]].tolist(), positives[j], kmax)
            rec = np.array([recall_at_k(r, k, len(positives[j])) for k in ks])
            nd = np.array([ndcg_at_k(r, k) for k in ks])
        result["recall"] += rec / n
        result["ndcg"] += nd / n
    return result
