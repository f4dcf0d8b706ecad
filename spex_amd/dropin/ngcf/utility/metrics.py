"""Ranking metrics of NGCF_SPEX/code/utility/metrics.py — the same definitions as the LightGCN copy
(recall_at_k :75-77, dcg_at_k :43-57, ndcg_at_k :60-70, precision_at_k :9-19, hit_at_k :80-85, F1 :87-91), shared with it."""
from spex_amd.dropin.utility1.metrics import (F1, accumulate_rank_metrics, dcg_at_k, hit_at_k, ndcg_at_k,  # noqa: F401
                                              precision_at_k, rank_metrics_batch, ranked_relevance, recall_at_k)


def auc(ground_truth, prediction):
    """metrics.py:93-98 (only reached with --test_flag full)."""
    try:
        from sklearn.metrics import roc_auc_score
        return roc_auc_score(y_true=ground_truth, y_score=prediction)
    except Exception:
        return 0.0
