"""Trust-task evaluation behind the reference's `utility2.batch_test_gnn` (trust_test5, :27-44): for each test path the
model's scores over all users are restricted to the path's candidate list (negatives then the true next user, last),
the top 50 taken, and recall/ndcg@{10,20,50} computed for the target's position — batched per slice instead of per
row.

Ranking is done where the reference does it: the slice's [B, 500] candidate scores (gathered on the device) are moved
to the host and ranked by the same torch.topk.  That matters because the reference's own candidate lists repeat the
target in ~15 % of the test paths (Data_process/path/data_process_path.py:199-204 looks the last user's friends up
with an int key in a dict keyed by strings, so nothing is excluded from the negative pool): the two copies tie exactly,
only the last one counts as the hit (:36-37), and which copy ranks first is topk's tie order — which differs between
the CPU and the GPU implementation and moves NDCG@10 by 0.014 on Epinion2."""
import numpy as np
import torch

import utility1.metrics as metrics

Ks = [10, 20, 50]


def get_performance(pos_item, r):
    return {"precision": np.array([metrics.precision_at_k(r, K) for K in Ks]),
            "recall": np.array([metrics.recall_at_k(r, K, len(pos_item)) for K in Ks]),
            "ndcg": np.array([metrics.ndcg_at_k(r, K) for K in Ks]),
            "hit_ratio": np.array([metrics.hit_at_k(r, K) for K in Ks])}


def test_one_user(pos_item, K_max_item):
    r = np.int32((np.asarray(K_max_item) - np.array(pos_item * 50)) == 0)
    return get_performance(pos_item, r)


def trust_test5(model, test_data):
    total = np.zeros(6)
    with torch.no_grad():
        for slice_indices in test_data.generate_batch(model.batch_size):
            scores, cand = model(None, None, None, slice_indices, test_data, 2)
            top = torch.gather(scores, 1, cand.to(scores.device)).cpu().topk(50, dim=1)[1].numpy()
            rel = (top == cand.shape[1] - 1).astype(np.float64)                 # the target is the last candidate
            rec, ndcg = metrics.rank_metrics_batch(rel, Ks, np.ones(len(rel)))
            total[:3] += rec.sum(0)
            total[3:] += ndcg.sum(0)
    return tuple(total / test_data.length)
