"""Dataset fixtures and synthetic graphs (inputs only — no arithmetic of the hot path lives here).

* Epinion2 is the only dataset whose raw data ships with the reference; tests/golden/epinion2_dataset.npz is its
  output of the reference preprocessing (see oracle/gen_golden.py, stage `mint`), and `materialise_rating_files`
  writes it back out in the reference's on-disk format (`<ds>.train.rating`, `<ds>.test.rating`,
  `<ds>.test.negative`; Data_process/rec/data_process_rec.py:401-416) so the Loader reads it like any dataset.
* Weibo / Twitter raw data are not in the reference (Google-Drive only, README.md:81): `synthetic_interactions`
  produces graphs of their published user counts (Trust_SPEX/code/main_trust.py:41-44) with a heavy-tailed degree
  law, and `scaled_interactions` the large graphs used for the HBM-roofline measurement.
"""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def xavier_uniform_np(rows, dim, rng):
    """U(-b, b), b = sqrt(6 / (rows + dim)) — nn.init.xavier_uniform_(gain=1) bounds (model.py:34-35) from a NumPy
    generator, so fixtures can name the initial tables by seed instead of storing 4 MB."""
    b = np.sqrt(6.0 / (rows + dim))
    return rng.uniform(-b, b, size=(rows, dim)).astype(np.float32)


def epinion2_tables(n_user_rows, m_item, dim=64, seed=2020):
    rng = np.random.default_rng(seed)
    return xavier_uniform_np(n_user_rows, dim, rng), xavier_uniform_np(m_item, dim, rng)


def load_epinion2(path=None):
    d = np.load(path or os.path.join(GOLDEN_DIR, "epinion2_dataset.npz"))
    return {k: d[k].astype(np.int64) for k in d.files}


def materialise_rating_files(root, name, train, test_users, test_pos, test_neg):
    """Write `<root>/<name>/rec/<name>.{train.rating,test.rating,test.negative}`; returns the data root (with a
    trailing slash) to pass as `--data_path`."""
    rec = os.path.join(root, name, "rec")
    os.makedirs(rec, exist_ok=True)
    train = np.asarray(train, np.int64)
    np.savetxt(os.path.join(rec, f"{name}.train.rating"), np.c_[train, np.ones(len(train), np.int64)], fmt="%d")
    test_users, test_pos, test_neg = (np.asarray(a, np.int64) for a in (test_users, test_pos, test_neg))
    with open(os.path.join(rec, f"{name}.test.rating"), "w") as fr, \
            open(os.path.join(rec, f"{name}.test.negative"), "w") as fn:
        for u, p, negs in zip(test_users, test_pos, test_neg):
            for j in negs:                       # negatives first, then the held-out positive (split(), :246-254)
                fr.write(f"{u} {j} 0\n")
            fr.write(f"{u} {p} 1\n")
            fn.write(str(u) + "".join(f" {j}" for j in negs) + "\n")
    return os.path.join(root, "")


def materialise_epinion2(root, name="epinion2"):
    d = load_epinion2()
    return materialise_rating_files(root, name, d["train"], d["test_users"], d["test_pos"], d["test_neg"])


# ------------------------------------------------------------------------------------------------ synthetic graphs
def synthetic_interactions(n_users, n_items, n_edges, seed=2020, device="cpu", zipf_a=1.1, sigma=1.0):
    """Distinct (user, item) pairs: users drawn with log-normal activity, items with Zipf(zipf_a) popularity.
    Returns int64 tensors (u, i) on `device`, sorted by (u, i); slightly fewer than n_edges after de-duplication."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    wu = torch.exp(sigma * torch.randn(n_users, generator=g, device=device, dtype=torch.float64))
    wi = 1.0 / torch.arange(1, n_items + 1, device=device, dtype=torch.float64).pow(zipf_a)
    wi = wi[torch.randperm(n_items, generator=g, device=device)]
    cu, ci = torch.cumsum(wu, 0), torch.cumsum(wi, 0)
    ru = torch.rand(n_edges, generator=g, device=device, dtype=torch.float64) * cu[-1]
    ri = torch.rand(n_edges, generator=g, device=device, dtype=torch.float64) * ci[-1]
    u = torch.searchsorted(cu, ru).clamp_(max=n_users - 1)
    i = torch.searchsorted(ci, ri).clamp_(max=n_items - 1)
    del ru, ri
    key = torch.unique(u * n_items + i)  # sorted
    # every user and the last item appear at least once so n_user / m_item are what was asked for
    extra = torch.cat([torch.arange(n_users, device=device) * n_items, torch.tensor([n_items - 1], device=device)])
    key = torch.unique(torch.cat([key, extra]))
    return key // n_items, key % n_items


def normalised_adjacency_torch(u, i, n_user_rows, n_items):
    """Same matrix as spex_amd.graph.lightgcn_norm_adj, built with torch ops on u.device (used for the large
    synthetic graphs, where NumPy's sort would dominate the benchmark's set-up time).  Returns NumPy CSR."""
    n = n_user_rows + n_items
    rows = torch.cat([u, i + n_user_rows])
    cols = torch.cat([i + n_user_rows, u])
    order = torch.argsort(rows * n + cols)
    rows, cols = rows[order], cols[order]
    del order
    deg = torch.bincount(rows, minlength=n)
    d_inv = deg.to(torch.float32).pow(-0.5)
    d_inv[torch.isinf(d_inv)] = 0.0
    val = (d_inv[rows] * 1.0) * d_inv[cols]
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=u.device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    return (rowptr.to(torch.int32).cpu().numpy(), cols.to(torch.int32).cpu().numpy(), val.cpu().numpy())


def epinion2_replicated(k, seed=2020, device="cpu", train=None):
    """"Epinion2 x K" (SURVEY.md 8d): K replicas of the Epinion2 users and items; the K copies of every interaction
    connect user replica j to item replica pi(j), pi a random permutation drawn per interaction.  Every replica of a
    user / item keeps exactly the degree of the original, so the graph has Epinion2's degree law (mean 26.8 stored
    entries per row, median 13, max 1 020) at any size, while its gathers are spread over the whole K-times larger
    embedding table.  Returns int64 tensors (u, i) and (n_user, m_item)."""
    if train is None:
        train = load_epinion2()["train"]
    tr = torch.as_tensor(np.asarray(train, np.int64), device=device)
    n_user, m_item = int(tr[:, 0].max()) + 1, int(tr[:, 1].max()) + 1
    if k == 1:
        return tr[:, 0].clone(), tr[:, 1].clone(), n_user, m_item
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    E = tr.shape[0]
    perm = torch.rand(E, k, generator=g, device=device).argsort(dim=1)          # [E, k] item replica per user replica
    rep = torch.arange(k, device=device)
    u = (tr[:, 0:1] + rep[None, :] * n_user).reshape(-1)
    i = (tr[:, 1:2] + perm * m_item).reshape(-1)
    return u, i, n_user * k, m_item * k


def scaled_graph(log2_nodes, seed=2020, device="cpu"):
    """The HBM-roofline workload: Epinion2 x K with K chosen so that the graph has about 2^log2_nodes nodes.
    Returns (rowptr, col, val, n_user_rows) as NumPy CSR."""
    k = max(1, round((1 << log2_nodes) / 15593))
    u, i, n_user, m_item = epinion2_replicated(k, seed=seed, device=device)
    rowptr, col, val = normalised_adjacency_torch(u, i, n_user + 1, m_item)
    return rowptr, col, val, n_user + 1
