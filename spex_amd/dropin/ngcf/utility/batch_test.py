"""Evaluation behind the reference's `utility.batch_test` surface (NGCF_SPEX/code/utility/batch_test.py).

`test(model, users_to_test)` / `rec_test(...)` (:19-32) run the model once at flag=1 and hand the two propagated
tables to `test_torch` (:119-172), which in the reference multiplies every 512-user block against ALL items
(`U I^T`, :158), copies the block to the host and ranks each user's 99 negatives + held-out items in a process pool.
Only those ~100 scores per user are ever read, so here a block is ONE scoring launch over its (user, candidate)
pairs on the device tables (spex_score_bce_f32, row stride = the concatenated width) followed by the reference's
ranking rule on the host (dict semantics for repeated candidates, heapq ties in insertion order) — no dense
[512, n_items] product, no pool.  recall/ndcg are accumulated user by user divided by the number of test users, as
:166-168 does.

Module-level names the drivers import — `args`, `Ks`, `data_generator`, `USR_NUM`, `ITEM_NUM`, `N_TRAIN`, `N_TEST`,
`NEG_ITEM`, `BATCH_SIZE` (:9-16) — are created on first use (the reference builds them at import time).
"""
import numpy as np
import torch

from spex_amd.dropin.ngcf.ngcf_parser import parse_known
from spex_amd.dropin.ngcf.utility import metrics
from spex_amd.dropin.ngcf.utility.load_data import Data

args = parse_known()
Ks = eval(args.Ks)
BATCH_SIZE = args.batch_size
_lazy = {}


def _singletons():
    if not _lazy:
        dg = Data(path=args.data_path + args.dataset, batch_size=args.batch_size)
        _lazy.update(data_generator=dg, USR_NUM=dg.n_users, ITEM_NUM=dg.n_items, N_TRAIN=dg.n_train, N_TEST=dg.n_test,
                     NEG_ITEM=dg.neg_item)
    return _lazy


def __getattr__(name):          # `from utility.batch_test import data_generator` etc.
    if name in ("data_generator", "USR_NUM", "ITEM_NUM", "N_TRAIN", "N_TEST", "NEG_ITEM"):
        return _singletons()[name]
    raise AttributeError(name)


def use_data(data):
    """Bind the module's dataset singleton to an existing Data object (tests, programmatic use)."""
    _lazy.clear()
    _lazy.update(data_generator=data, USR_NUM=data.n_users, ITEM_NUM=data.n_items, N_TRAIN=data.n_train, N_TEST=data.n_test,
                 NEG_ITEM=data.neg_item)


def ranklist_by_heapq(user_pos_test, test_items, rating, Ks):
    r = metrics.ranked_relevance(test_items, [rating[i] for i in test_items], user_pos_test, max(Ks))
    return r, 0.0


def get_performance(user_pos_test, r, auc, Ks):
    return {"recall": np.array([metrics.recall_at_k(r, K, len(user_pos_test)) for K in Ks]),
            "ndcg": np.array([metrics.ndcg_at_k(r, K) for K in Ks])}


def test_one_user(x):
    """x = (rating row over all items, uid) — :91-116."""
    rating, u = x[0], x[1]
    dg = _singletons()["data_generator"]
    user_pos_test = dg.test_set[u]
    test_items = dg.neg_item[u] + user_pos_test
    r, auc = ranklist_by_heapq(user_pos_test, test_items, rating, Ks)
    return get_performance(user_pos_test, r, auc, Ks)


def _scores(ua, ia, users, items):
    """<ua[u], ia[i]> for the listed pairs: one libspexhip launch on the device tables (no host path)."""
    if not (ua.is_cuda and ia.is_cuda):
        raise RuntimeError("utility.batch_test scores on the GPU: pass the model's device tables (no CPU fallback)")
    from spex_amd import ops
    gamma, _ = ops.score_bce(ua.detach(), ia.detach(), torch.from_numpy(users), torch.from_numpy(items))
    return gamma.cpu().numpy()


def test_torch(ua_embeddings, ia_embeddings, users_to_test, drop_flag=False, batch_test_flag=False):
    dg = _singletons()["data_generator"]
    test_users = list(users_to_test)
    n_test_users = len(test_users)
    result = {"recall": np.zeros(len(Ks)), "ndcg": np.zeros(len(Ks))}
    if n_test_users == 0:
        return result
    ua = ua_embeddings if ua_embeddings.is_contiguous() else ua_embeddings.contiguous()
    ia = ia_embeddings if ia_embeddings.is_contiguous() else ia_embeddings.contiguous()
    u_batch_size = BATCH_SIZE * 2
    cand_all, pos_all, score_all = [], [], []
    for start in range(0, n_test_users, u_batch_size):          # the reference's 512-user blocks (:131-136)
        block = test_users[start:start + u_batch_size]
        cand = [list(dg.neg_item[u]) + list(dg.test_set[u]) for u in block]
        lens = [len(c) for c in cand]
        flat_items = np.fromiter((i for c in cand for i in c), np.int64, sum(lens))
        flat_users = np.repeat(np.asarray(block, np.int64), lens)
        if flat_items.size and (flat_items.max() >= ia.shape[0] or flat_users.max() >= ua.shape[0]):
            raise IndexError("test candidate outside the embedding tables")        # the dense product would raise too
        score_all.append(_scores(ua, ia, flat_users, flat_items))
        cand_all.extend(cand)
        pos_all.extend(dg.test_set[u] for u in block)
    out = metrics.accumulate_rank_metrics(cand_all, np.concatenate(score_all), pos_all, Ks)
    result["recall"] += out["recall"]
    result["ndcg"] += out["ndcg"]
    return result


def test(model, users_to_test, drop_flag=False, batch_test_flag=False):
    model.eval()
    with torch.no_grad():
        ua_embeddings, ia_embeddings = model(None, None, None, 1)
    return test_torch(ua_embeddings, ia_embeddings, users_to_test)


def rec_test(model, users_to_test, drop_flag=False, batch_test_flag=False):
    model.eval()
    with torch.no_grad():
        ua_embeddings, ia_embeddings = model(None, None, None, None, None, 1)
    return test_torch(ua_embeddings, ia_embeddings, users_to_test)
