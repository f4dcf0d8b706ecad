"""Pin the CPU oracle against vectors minted from the reference itself (oracle/gen_golden.py).

These are the golden rows G1..G10 of SURVEY.md 8c.  The oracle is test infrastructure; once it agrees with the
reference here, the GPU parity tests (tests/test_gpu_*.py) compare the HIP path against it and against the same
goldens.
"""
import hashlib

import numpy as np
import pytest


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ---------------------------------------------------------------- G1 adjacency
def test_g1_adjacency_tiny_bit_exact(oracle, golden):
    g = golden("lightgcn_tiny")
    tp = g["train_pairs"]
    rowptr, col, val = oracle.build_norm_adj_lightgcn(tp[:, 0], tp[:, 1], int(g["n_user"]), int(g["m_item"]))
    assert np.array_equal(rowptr, g["rowptr"]) and np.array_equal(col, g["col"])
    assert np.array_equal(val, g["val"])          # bit-for-bit


def test_g1_adjacency_epinion2_hashes(oracle, golden, epinion2):
    g = golden("lightgcn_epinion2")
    tr = epinion2["train"]
    rowptr, col, val = oracle.build_norm_adj_lightgcn(tr[:, 0], tr[:, 1], int(g["n_user"]), int(g["m_item"]))
    assert len(col) == int(g["nnz"]) == 418608 and len(rowptr) - 1 == 15593
    assert sha(rowptr) == str(g["rowptr_sha"]) and sha(col) == str(g["col_sha"]) and sha(val) == str(g["val_sha"])
    e = g["edge_idx"]
    assert np.array_equal(col[e], g["edge_col"]) and np.array_equal(val[e], g["edge_val"])
    # symmetric, as the backward relies on
    t = oracle.csr_transpose(rowptr, col, val, len(rowptr) - 1)
    assert np.array_equal(t[0], rowptr) and np.array_equal(t[1], col) and np.array_equal(t[2], val)


def test_g10_ngcf_adjacency(oracle, golden, epinion2):
    g = golden("ngcf_tiny")
    tp = g["train_pairs"]
    rowptr, col, val = oracle.build_norm_adj_ngcf(tp[:, 0], tp[:, 1], int(g["n_users"]), int(g["n_items"]))
    assert np.array_equal(rowptr, g["rowptr"]) and np.array_equal(col, g["col"]) and np.array_equal(val, g["val"])
    g = golden("ngcf_epinion2")
    tr = epinion2["train"]
    rowptr, col, val = oracle.build_norm_adj_ngcf(tr[:, 0], tr[:, 1], int(g["n_users"]), int(g["n_items"]))
    assert len(col) == int(g["nnz"]) == 434200
    assert sha(rowptr) == str(g["rowptr_sha"]) and sha(col) == str(g["col_sha"]) and sha(val) == str(g["val_sha"])


# ---------------------------------------------------------------- G2 propagation
def _epinion2_setup(oracle, golden, epinion2):
    from spex_amd.datasets import epinion2_tables
    g = golden("lightgcn_epinion2")
    tr = epinion2["train"]
    nu, mi = int(g["n_user"]), int(g["m_item"])
    csr = oracle.build_norm_adj_lightgcn(tr[:, 0], tr[:, 1], nu, mi)
    uw, iw = epinion2_tables(nu + 1, mi)
    return g, csr, np.concatenate([uw, iw]), nu + 1


def test_g2_propagation_tiny_bit_exact(oracle, golden):
    g = golden("lightgcn_tiny")
    csr = (g["rowptr"], g["col"], g["val"])
    out, layers = oracle.propagate_mean(*csr, g["E0"], 3, return_layers=True)
    for l in range(3):
        assert np.array_equal(layers[l], g[f"E{l + 1}"])   # same fmaf chain as ATen's CPU kernel
    assert np.array_equal(out, g["light_out"])


def test_g2_propagation_epinion2(oracle, golden, epinion2):
    g, csr, E0, _ = _epinion2_setup(oracle, golden, epinion2)
    rows = g["sample_rows"]
    assert np.array_equal(E0[rows], g["E0_rows"])          # the seeded initialiser reproduces the fixture's E0
    out, layers = oracle.propagate_mean(*csr, E0, 3, return_layers=True, n_threads=8)
    for l in range(3):
        assert rel_err(layers[l][rows], g[f"E{l + 1}_rows"]) <= 1e-6
        assert np.allclose(layers[l].astype(np.float64).sum(0), g[f"E{l + 1}_colsum"], rtol=1e-6, atol=1e-7)
        assert np.isclose(np.sqrt((layers[l].astype(np.float64) ** 2).sum()), g[f"E{l + 1}_fro"], rtol=1e-7)
    assert rel_err(out[rows], g["light_out_rows"]) <= 1e-6
    assert np.allclose(out.astype(np.float64).sum(0), g["light_out_colsum"], rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------- G3 scoring, loss, gradient through the propagation
@pytest.mark.parametrize("ds", ["tiny", "epinion2"])
def test_g3_loss_and_grad(oracle, golden, epinion2, ds):
    if ds == "tiny":
        g = golden("lightgcn_tiny")
        csr, E0, n_u = (g["rowptr"], g["col"], g["val"]), g["E0"], int(g["n_user"]) + 1
    else:
        g, csr, E0, n_u = _epinion2_setup(oracle, golden, epinion2)
    u, i, y = g["batch_users"][0], g["batch_items"][0], g["batch_labels"][0]
    gamma, loss, G = oracle.lightgcn_loss_and_grad(*csr, E0, n_u, 3, u, i, y.astype(np.float32), n_threads=8)
    assert rel_err(gamma, g["g3_gamma"]) <= 2e-6
    assert abs(float(loss) - float(g["g3_loss"])) <= 1e-6
    if ds == "tiny":
        assert rel_err(G, g["g3_grad"]) <= 1e-5
    else:
        assert rel_err(G[g["sample_rows"]], g["g3_grad_rows"]) <= 1e-5
    assert np.allclose(G.astype(np.float64).sum(0), g["g3_grad_colsum"], rtol=1e-4, atol=1e-9)
    assert np.isclose(np.sqrt((G.astype(np.float64) ** 2).sum()), g["g3_grad_fro"], rtol=1e-5)


@pytest.mark.parametrize("L", [2, 4])
def test_g3_first_step_loss_at_other_depths(oracle, golden, epinion2, L):
    """The oracle at the reference's other depths (`main_rec.py --layer 2` / `--layer 4`; goldens lightgcn_epinion2_L{2,4}.npz, minted by
    oracle/gen_golden.py --stage epochs-L{2,4}-epinion2): the first training step's mean BCE loss — the seeded initial tables, the
    reference's own first batch, L layers of utility1/model.py:83-97 (the later steps need the run's batches: the GPU replay,
    tests/test_gpu_dropin.py: test_two_and_four_layer_runs_match_the_reference, reproduces all 120)."""
    import torch
    _, csr, _, n_u = _epinion2_setup(oracle, golden, epinion2)
    g = golden(f"lightgcn_epinion2_L{L}")
    # the run's initial tables: main_rec.py:15,22 — set_seed, then model.py:32-35 (two nn.Embedding, xavier_uniform_ on each)
    torch.manual_seed(int(g["seed"]))
    eu, ei = torch.nn.Embedding(n_u, 64), torch.nn.Embedding(len(csr[0]) - 1 - n_u, 64)
    torch.nn.init.xavier_uniform_(eu.weight, gain=1)
    torch.nn.init.xavier_uniform_(ei.weight, gain=1)
    E0 = torch.cat([eu.weight, ei.weight]).detach().numpy()
    u, i, y = (g["first_batch"][k] for k in range(3))
    _, loss, G = oracle.lightgcn_loss_and_grad(*csr, E0, n_u, L, u.astype(np.int64), i.astype(np.int64), y.astype(np.float32), n_threads=8)
    assert abs(float(loss) - float(g["step_losses"][0])) <= 2e-6, (L, float(loss), float(g["step_losses"][0]))
    assert np.isfinite(G).all() and np.abs(G).max() > 0


# ---------------------------------------------------------------- G4 Adam
def test_g4_adam_tiny(oracle, golden):
    g = golden("lightgcn_tiny")
    csr, n_u = (g["rowptr"], g["col"], g["val"]), int(g["n_user"]) + 1
    W = g["E0"].copy()
    m, v = np.zeros_like(W), np.zeros_like(W)
    for s in range(5):
        u, i, y = g["batch_users"][s], g["batch_items"][s], g["batch_labels"][s]
        _, loss, G = oracle.lightgcn_loss_and_grad(*csr, W, n_u, 3, u, i, y.astype(np.float32))
        assert abs(float(loss) - float(g["g4_losses"][s])) <= 2e-6
        oracle.adam_step(W, G, m, v, s + 1)
        if s + 1 in (1, 2, 5):
            assert rel_err(W, g[f"g4_w_step{s + 1}"]) <= 2e-6


# ---------------------------------------------------------------- G5 metrics
def test_g5_metric_cases(oracle, golden):
    g = golden("g5_metric_cases")
    for k in range(int(g["n_cases"])):
        items, scores = g[f"items_{k}"], g[f"scores_{k}"]
        pos = [int(items[-1])]
        r = oracle.ranklist(items, scores, pos)
        assert r == [int(x) for x in g[f"r_{k}"]]
        assert np.allclose([oracle.recall_at_k(r, K, 1) for K in oracle.KS], g[f"recall_{k}"], atol=0)
        assert np.allclose([oracle.ndcg_at_k(r, K) for K in oracle.KS], g[f"ndcg_{k}"], atol=1e-15)


def _adam5(oracle, g, csr, W0, n_u):
    W = W0.copy()
    m, v = np.zeros_like(W), np.zeros_like(W)
    for s in range(5):
        u, i, y = g["batch_users"][s], g["batch_items"][s], g["batch_labels"][s]
        _, _, G = oracle.lightgcn_loss_and_grad(*csr, W, n_u, 3, u, i, y.astype(np.float32), n_threads=8)
        oracle.adam_step(W, G, m, v, s + 1)
    return W


@pytest.mark.parametrize("ds", ["tiny", "epinion2"])
def test_g5_end_to_end(oracle, golden, epinion2, ds):
    """test() (batch_test.py:12-40) on the tables after the five Adam steps of G4: HR@K / NDCG@K gate (<= 1e-4)."""
    if ds == "tiny":
        g = golden("lightgcn_tiny")
        csr, W0, n_u = (g["rowptr"], g["col"], g["val"]), g["E0"], int(g["n_user"]) + 1
        tu, tp, tn = g["test_users"], g["test_pos"], g["test_neg"]
    else:
        g, csr, W0, n_u = _epinion2_setup(oracle, golden, epinion2)
        tu, tp, tn = epinion2["test_users"], epinion2["test_pos"], epinion2["test_neg"]
    W = _adam5(oracle, g, csr, W0, n_u)
    out = oracle.propagate_mean(*csr, W, 3, n_threads=8)
    ratings = {int(u): [int(p)] for u, p in zip(tu, tp)}
    negatives = {int(u): [int(x) for x in n] for u, n in zip(tu, tn)}

    def score(u, items):
        return oracle.score_bce(out[:n_u], out[n_u:], np.full(len(items), u), np.asarray(items))

    for k, u in enumerate(g["g5_users"][:8]):
        items = negatives[int(u)] + ratings[int(u)]
        assert rel_err(score(int(u), items), g["g5_scores"][k]) <= 1e-4
    res = oracle.evaluate(score, ratings, negatives)
    assert np.abs(res["recall"] - g["g5_recall"]).max() <= 1e-4
    assert np.abs(res["ndcg"] - g["g5_ndcg"]).max() <= 1e-4


# ---------------------------------------------------------------- G6 sampler replay
def test_g6_sampler_replay(oracle, golden):
    g = golden("lightgcn_tiny")
    tp = g["train_pairs"]
    train_set = set(map(tuple, tp.tolist()))
    np.random.seed(2020)
    neg = oracle.ng_sample_replay(tp.tolist(), int(g["m_item"]), train_set)
    assert np.array_equal(np.asarray(neg), g["g6_neg"])


# ---------------------------------------------------------------- G7 NGCF forward
@pytest.mark.parametrize("ds", ["tiny", "epinion2"])
def test_g7_ngcf_forward(oracle, golden, epinion2, ds):
    g = golden(f"ngcf_{ds}")
    if ds == "tiny":
        csr, uw, iw = (g["rowptr"], g["col"], g["val"]), g["user_w"], g["item_w"]
    else:
        from spex_amd.datasets import epinion2_tables
        tr = epinion2["train"]
        csr = oracle.build_norm_adj_ngcf(tr[:, 0], tr[:, 1], int(g["n_users"]), int(g["n_items"]))
        uw, iw = epinion2_tables(int(g["n_users"]) + 1, int(g["n_items"]))
    out = oracle.ngcf_forward(*csr, uw, iw, g["W_gc"], g["b_gc"], g["W_bi"], g["b_bi"])
    ref = g["all_emb"] if ds == "tiny" else g["all_emb_rows"]
    got = out if ds == "tiny" else out[g["sample_rows"]]
    assert rel_err(got, ref) <= 1e-5
    assert np.allclose(out.astype(np.float64).sum(0), g["all_emb_colsum"], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- G8 expert gate
def test_g8_expert_gate(oracle, golden):
    g = golden("lightgcn_tiny")
    csr, n_u = (g["rowptr"], g["col"], g["val"]), int(g["n_user"]) + 1
    E0 = g["E0"]
    out = oracle.propagate_mean(*csr, E0, 3)
    mu = oracle.expert_gate(E0[:n_u], out[:n_u], g["g8_att_exp1"])
    mi = oracle.expert_gate(E0[n_u:], out[n_u:], g["g8_att_exp2"])
    gamma = oracle.score_bce(mu, mi, g["batch_users"][0], g["batch_items"][0])
    assert rel_err(gamma, g["g8_gamma"]) <= 1e-5


# ---------------------------------------------------------------- G9 dropout with the reference's injected mask
def test_g9_dropout_tiny(oracle, golden):
    g = golden("lightgcn_tiny")
    csr = (g["rowptr"], g["col"], g["val"])
    keep = oracle.dropout_keep_mask(g["g9_rand"], float(g["g9_keep"]))
    out = oracle.propagate_mean_masked(*csr, keep, float(g["g9_keep"]), g["E0"], 3)
    assert rel_err(out, g["g9_light_out"]) <= 1e-6


def test_bpr_closed_form_self_consistency(oracle):
    """bpr_* has no reference counterpart (parity unpinned): check the C closed form against NumPy fp64."""
    rng = np.random.default_rng(0)
    U, I = rng.normal(size=(30, 64)).astype(np.float32), rng.normal(size=(50, 64)).astype(np.float32)
    u, p, n = rng.integers(0, 30, 200), rng.integers(0, 50, 200), rng.integers(0, 50, 200)
    loss, Un, In = oracle.bpr_sgd(U, I, U, I, u, p, n, lr=0.1, reg=0.0)
    x = (U[u].astype(np.float64) * (I[n].astype(np.float64) - I[p])).sum(1)
    assert np.isclose(loss, np.mean(np.logaddexp(0, x)))
    s = 1 / (1 + np.exp(-x)) / 200
    Ue = U.astype(np.float64).copy()
    np.add.at(Ue, u, -0.1 * s[:, None] * (I[n].astype(np.float64) - I[p]))
    assert np.allclose(Un, Ue)


def test_learned_edge_value_restatements_vs_torch_sparse(oracle):
    """SURVEY.md 8f #3 (Diffnet++): the reference ops are TensorFlow's (absent here) — parity unpinned; the numpy
    restatements are at least held against torch's independent CPU implementations of the same documented ops
    (torch.sparse.softmax, autograd of torch.sparse.mm w.r.t. the values)."""
    import torch
    rng = np.random.default_rng(11)
    n_rows, n_cols, d = 60, 45, 16
    deg = rng.integers(0, 12, n_rows)
    deg[3] = 0
    deg[7] = n_cols
    rowptr = np.zeros(n_rows + 1, np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = np.concatenate([np.sort(rng.choice(n_cols, k, replace=False)) for k in deg]).astype(np.int64)
    rows = np.repeat(np.arange(n_rows), deg)
    v = rng.normal(size=len(col))
    idx = torch.from_numpy(np.stack([rows, col]))
    tv = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    S = torch.sparse_coo_tensor(idx, tv, (n_rows, n_cols))
    y = torch.sparse.softmax(S, dim=1).coalesce().values()
    got = oracle.edge_softmax(rowptr, v, np.float64)
    assert np.allclose(got, y.detach().numpy(), rtol=1e-12, atol=1e-14)
    gy = rng.normal(size=len(col))
    (y * torch.from_numpy(gy)).sum().backward()
    assert np.allclose(oracle.edge_softmax_bwd(rowptr, got, gy, np.float64), tv.grad.numpy(), rtol=1e-10, atol=1e-13)
    # SDDMM == d/dval of sum(G * (A(val) X))
    X = rng.normal(size=(n_cols, d))
    G = rng.normal(size=(n_rows, d))
    tv2 = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    Y = torch.sparse.mm(torch.sparse_coo_tensor(idx, tv2, (n_rows, n_cols)), torch.from_numpy(X))
    (Y * torch.from_numpy(G)).sum().backward()
    assert np.allclose(oracle.sddmm(rowptr, col, G, X, np.float64), tv2.grad.numpy(), rtol=1e-12, atol=1e-13)


def test_g11_dual_task_restatement_vs_the_reference(golden):
    """oracle/trust_oracle.py: dual_task_losses — the whole dual-task forward of model_expert_s.py restated in torch fp64 on the
    CPU — against the reference's own run of the same model (G11, trust_tiny.npz: parameters by value, a rec batch, 30 trust
    paths): both losses and EVERY parameter's gradient of loss1 + loss2.  This pins the restatement that the config-5 GPU tests
    on the Weibo-shaped graph (where the reference itself cannot run: its data are not shipped) are held against."""
    import torch
    from oracle.trust_oracle import dual_task_losses
    g, lg = golden("trust_tiny"), golden("lightgcn_tiny")
    P = {k[6:].replace("__", "."): torch.from_numpy(g[k]).double().requires_grad_(True) for k in g.files
         if k.startswith("state_") and k != "state_task_weights"}
    sl = g["slice_indices"]
    loss1, loss2 = dual_task_losses(lg["rowptr"], lg["col"], lg["val"], P, lg["batch_users"][0], lg["batch_items"][0],
                                    lg["batch_labels"][0].astype(np.float64), g["train_inputs"][sl], g["train_mask"][sl],
                                    g["train_targets"][sl])
    assert abs(loss1.item() - float(g["loss1"])) <= 2e-6 and abs(loss2.item() - float(g["loss2"])) <= 2e-5
    (loss1 + loss2).backward()
    checked = 0
    for name, p in P.items():
        key = "grad_" + name.replace(".", "__")
        if key in g.files:
            want = g[key].reshape(p.shape)
            err = np.abs(p.grad.numpy() - want).max()
            # (att_t: a two-way softmax's parameter gradient has exactly antisymmetric columns; the reference's fp32 columns
            #  differ from that by 3e-7 of their own rounding, hence the absolute alternative)
            assert err <= 2e-5 * max(np.abs(want).max(), 1e-6) or err <= 1e-6, (name, err)
            checked += 1
    assert checked >= 16
