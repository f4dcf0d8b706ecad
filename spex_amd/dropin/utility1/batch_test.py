"""Evaluation loop behind the reference's `utility1.batch_test` surface (test / rec_test / test_one_user / ...).

Reference: LightGCN_SPEX/code/utility1/batch_test.py:12-90 scores each test user's 99 negatives + 1 positive with one
model forward per user — i.e. one full 3-layer propagation per user.  Here `test()` scores every user's 100 candidates
in ONE scoring launch on the cached propagated tables, then ranks on the host with the reference's exact tie rule
(heapq.nlargest over a dict == stable descending sort in insertion order; a repeated item keeps its first position
and its last score).  recall/ndcg@{10,20,50} accumulate in the same order and dtype as batch_test.py:19-24.
"""
import heapq

import numpy as np
import torch

import utility1.metrics as metrics
from lg_parser import build_parser

args, _ = build_parser().parse_known_args()
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
Ks = [10, 20, 50]
BATCH_SIZE = 256


def ranklist_by_heapq(user_pos_test, rating):
    top = heapq.nlargest(max(Ks), rating, key=rating.get)
    pos = set(user_pos_test)
    return [1 if i in pos else 0 for i in top]


def get_performance(user_pos_test, r):
    return {"recall": np.array([metrics.recall_at_k(r, K, len(user_pos_test)) for K in Ks]),
            "ndcg": np.array([metrics.ndcg_at_k(r, K) for K in Ks])}


def _call_model(model, users, items, dual):
    u, i = torch.from_numpy(users).long(), torch.from_numpy(items).long()
    if dual:
        return model(users=u, items=i, labels=None, slice_indices=None, trust_data=None, flag=1)
    return model(users=u, items=i, labels=None, flag=1)


def _one_user(user, test_item, neg_item, model, dual):
    test_items = list(neg_item) + list(test_item)
    users = np.full(len(test_items), user, dtype=np.int64)
    pred = _call_model(model, users, np.asarray(test_items, np.int64), dual).cpu().tolist()
    rating = {}
    for it, p in zip(test_items, pred):
        rating[it] = p
    return get_performance(test_item, ranklist_by_heapq(test_item, rating))


def test_one_user(user, test_item, neg_item, model):
    return _one_user(user, test_item, neg_item, model, dual=False)


def test_one_user_rec(user, test_item, neg_item, model):
    return _one_user(user, test_item, neg_item, model, dual=True)


def _test_all(model, testRatings, testNegatives, dual):
    result = {"recall": np.zeros(len(Ks)), "ndcg": np.zeros(len(Ks))}
    users = list(testRatings.keys())
    n_test_users = len(users)
    if n_test_users == 0:
        return result
    cand = [list(testNegatives[u]) + list(testRatings[u]) for u in users]
    lens = np.array([len(c) for c in cand])
    flat_items = np.concatenate([np.asarray(c, np.int64) for c in cand])
    flat_users = np.repeat(np.asarray(users, np.int64), lens)
    with torch.no_grad():
        scores = _call_model(model, flat_users, flat_items, dual).cpu().numpy().astype(np.float64)
    return metrics.accumulate_rank_metrics(cand, scores, [testRatings[u] for u in users], Ks)


def test(model, testRatings, testNegatives):
    return _test_all(model, testRatings, testNegatives, dual=False)


def rec_test(model, testRatings, testNegatives):
    return _test_all(model, testRatings, testNegatives, dual=True)
