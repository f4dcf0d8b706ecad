"""Host-side logic of the drop-in surface, on CPU: flags, file formats, adjacency construction, negative sampler,
ranking metrics, the evaluation loop's tie semantics, the launcher's import order."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO
from test_oracle_golden import sha


def test_lg_parser_defaults_match_reference_flags():
    import lg_parser
    a = lg_parser.parse_args_r([])
    # lg_parser.py:5-21 of the reference
    assert (a.cuda_id, a.data_path, a.dataset, a.nb_heads, a.recdim, a.layer) == ("0", "../data/", "twitter", 3, 64, 3)
    assert (a.lr, a.dropout, a.keepprob, a.a_fold, a.epochs, a.seed) == (0.001, 0, 0.6, 100, 50, 2020)
    assert (a.A_split, a.batch_size, a.batchSize, a.hiddenSize, a.nonhybrid, a.act) == (0, 256, 256, 64, False, 1)
    b = lg_parser.parse_args_r("--dataset epinion2 --recdim 32 --layer 2 --dropout 1 --keepprob 0.3 --nonhybrid".split())
    assert (b.dataset, b.recdim, b.layer, b.dropout, b.keepprob, b.nonhybrid) == ("epinion2", 32, 2, 1, 0.3, True)


@pytest.fixture(scope="module")
def tiny_loader(tmp_path_factory, golden):
    from spex_amd.datasets import materialise_rating_files
    import lg_parser
    import utility1.dataloader as dl
    g = golden("lightgcn_tiny")
    root = materialise_rating_files(str(tmp_path_factory.mktemp("data")), "tiny", g["train_pairs"], g["test_users"],
                                    g["test_pos"], g["test_neg"])
    args = lg_parser.parse_args_r(["--dataset", "tiny", "--data_path", root])
    return dl.Loader(args), g


def test_loader_surface_and_adjacency_tiny(tiny_loader):
    ld, g = tiny_loader
    assert (ld.n_user, ld.m_item, ld.n_users, ld.m_items) == (50, 60, 50, 60)
    assert ld.rec_train_data == g["train_pairs"].tolist()
    assert ld.train_mat.shape == (51, 60) and ld.UserItemNet.shape == (51, 60)
    assert (0, int(g["train_pairs"][0, 1])) in ld.train_mat
    assert list(ld.testRatings.keys()) == g["test_users"].tolist()
    assert [ld.testRatings[u][0] for u in ld.testRatings] == g["test_pos"].tolist()
    assert [ld.testNegatives[u] for u in ld.testNegatives] == g["test_neg"].tolist()
    rowptr, col, val = ld.build_adjacency()
    assert np.array_equal(rowptr, g["rowptr"]) and np.array_equal(col, g["col"]) and np.array_equal(val, g["val"])


def test_adjacency_builders_epinion2_bit_exact(golden, epinion2):
    from spex_amd.graph import lightgcn_norm_adj, ngcf_norm_adj, csr_transpose
    g = golden("lightgcn_epinion2")
    tr = epinion2["train"]
    rowptr, col, val = lightgcn_norm_adj(tr[:, 0], tr[:, 1], int(g["n_user"]), int(g["m_item"]))
    assert (sha(rowptr), sha(col), sha(val)) == (str(g["rowptr_sha"]), str(g["col_sha"]), str(g["val_sha"]))
    t_rowptr, t_col, t_val, eid = csr_transpose(rowptr, col, val, len(rowptr) - 1)
    assert np.array_equal(t_rowptr, rowptr) and np.array_equal(t_col, col) and np.array_equal(t_val, val)
    assert np.array_equal(val[eid], t_val) and sorted(eid.tolist()) == list(range(len(col)))
    gn = golden("ngcf_epinion2")
    rowptr, col, val = ngcf_norm_adj(tr[:, 0], tr[:, 1], int(gn["n_users"]), int(gn["n_items"]))
    assert (sha(rowptr), sha(col), sha(val)) == (str(gn["rowptr_sha"]), str(gn["col_sha"]), str(gn["val_sha"]))


def test_adjacency_torch_builder_matches_numpy():
    from spex_amd.datasets import synthetic_interactions, normalised_adjacency_torch
    from spex_amd.graph import lightgcn_norm_adj
    u, i = synthetic_interactions(300, 900, 6000, seed=3)
    a = normalised_adjacency_torch(u, i, 301, 900)
    b = lightgcn_norm_adj(u.numpy(), i.numpy(), 300, 900)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.abs(a[2] - b[2]).max() <= 1e-7          # torch pow vs numpy pow may differ in the last bit


def test_sampler_replays_reference_stream_tiny(tiny_loader):
    import utility1.dataloader as dl
    ld, g = tiny_loader
    td = dl.LightTrainData(ld.rec_train_data, ld.m_item, ld.train_mat)
    np.random.seed(2020)
    td.ng_sample()
    assert np.array_equal(np.asarray(td.features_ng), g["g6_neg"])
    assert len(td) == int(g["g6_len"]) == 6 * len(ld.rec_train_data)
    assert list(td[0]) == g["g6_item0"].tolist() and list(td[len(td) - 1]) == g["g6_item_last"].tolist()
    # through torch's DataLoader (batched fetch + default collate) a batch unpacks like the reference's: three [B] int64
    from torch.utils.data import DataLoader
    want = [td[k] for k in range(len(td))]
    got = []
    for user, item, label in DataLoader(td, batch_size=7, shuffle=False):
        assert user.dtype == item.dtype == label.dtype == torch.int64 and user.dim() == 1
        got += list(zip(user.tolist(), item.tolist(), label.tolist()))
    assert got == [tuple(w) for w in want]


def test_sampler_replay_dense_case_against_oracle(oracle):
    """High rejection rate (30 % of all pairs are positives) and the RNG state afterwards must match too."""
    import scipy.sparse as sp
    import utility1.dataloader as dl
    rng = np.random.default_rng(1)
    U, I = 40, 25
    dense = rng.random((U, I)) < 0.3
    dense[:, 0] = True
    pairs = np.argwhere(dense)
    mat = sp.coo_matrix((np.ones(len(pairs), np.float32), (pairs[:, 0], pairs[:, 1])), shape=(U + 1, I)).todok()
    td = dl.LightTrainData(pairs.tolist(), I, mat)
    np.random.seed(7)
    td.ng_sample(block=13)                     # small blocks: the slot offset is carried across many block seams
    assert td.features_ng == oracle.ng_sample_replay.__globals__["np"].asarray(td.features_ng).tolist()
    first = td.features_ng
    np.random.seed(7)
    td.ng_sample()
    assert td.features_ng == first             # block size does not change the result
    after_fast = np.random.randint(1 << 30)
    np.random.seed(7)
    ref = oracle.ng_sample_replay(pairs.tolist(), I, set(map(tuple, pairs.tolist())))
    after_ref = np.random.randint(1 << 30)
    assert td.features_ng == ref
    assert after_fast == after_ref
    assert not any((u, j) in mat for u, j in td.features_ng)


def test_metrics_match_reference_cases(golden):
    import utility1.metrics as M
    g = golden("g5_metric_cases")
    rel, rec, ndcg = [], [], []
    for k in range(int(g["n_cases"])):
        r = [int(x) for x in g[f"r_{k}"]]
        assert np.allclose([M.recall_at_k(r, K, 1) for K in (10, 20, 50)], g[f"recall_{k}"], atol=0)
        assert np.allclose([M.ndcg_at_k(r, K) for K in (10, 20, 50)], g[f"ndcg_{k}"], atol=1e-15)
        rel.append(r + [0] * (50 - len(r))); rec.append(g[f"recall_{k}"]); ndcg.append(g[f"ndcg_{k}"])
    br, bn = M.rank_metrics_batch(np.asarray(rel), (10, 20, 50), np.ones(len(rel)))
    assert np.allclose(br, rec, atol=0) and np.allclose(bn, ndcg, atol=1e-15)


class _TableModel:
    """Stands in for the model in the evaluation loop: scores from a fixed table (host logic test)."""

    def __init__(self, table):
        self.table = table

    def __call__(self, users, items, labels, flag=1, **kw):
        return torch.from_numpy(self.table[users.numpy(), items.numpy()])


def test_eval_loop_tie_and_duplicate_semantics(oracle, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["x"])
    import utility1.batch_test as bt
    rng = np.random.default_rng(4)
    U, I = 30, 200
    table = np.round(rng.normal(size=(U, I)), 1).astype(np.float32)      # many ties
    ratings = {u: [int(rng.integers(I))] for u in range(U)}
    negs = {}
    for u in range(U):
        c = [int(x) for x in rng.permutation(I) if x != ratings[u][0]][:99]
        if u % 7 == 0:
            c[5] = c[3]                                                   # duplicated candidate -> dict semantics
        negs[u] = c
    got = bt.test(_TableModel(table), ratings, negs)
    want = oracle.evaluate(lambda u, items: table[u, items], ratings, negs)
    assert np.array_equal(got["recall"], want["recall"]) and np.allclose(got["ndcg"], want["ndcg"], atol=1e-15)
    one = bt.test_one_user(3, ratings[3], negs[3], _TableModel(table))
    r = oracle.ranklist(negs[3] + ratings[3], table[3, negs[3] + ratings[3]], ratings[3])
    assert np.allclose(one["ndcg"], [oracle.ndcg_at_k(r, k) for k in (10, 20, 50)])


def test_model_refuses_cpu(tiny_loader):
    """No CPU fallback: the model needs the HIP library's device graph."""
    import utility1.model as model
    import lg_parser
    ld, _ = tiny_loader
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        model.LightGCN(lg_parser.parse_args_r(["--dataset", "tiny"]), ld)


def test_launcher_puts_dropin_modules_first(tmp_path):
    script = tmp_path / "main_probe.py"
    script.write_text("import lg_parser, utility1.dataloader as d, utility.dataloader as d2, world\n"
                      "print(lg_parser.__file__); print(d.__file__); print(d2 is d); print(world.config['latent_dim_rec'])\n")
    out = subprocess.run([sys.executable, "-m", "spex_amd.dropin", str(script), "--recdim", "32"], cwd=REPO,
                         capture_output=True, text=True, check=True).stdout.split()
    assert out[0].endswith("spex_amd/dropin/lg_parser.py") and out[1].endswith("spex_amd/dropin/utility1/dataloader.py")
    assert out[2] == "True" and out[3] == "32"


def test_epinion2_fixture_round_trips_through_the_file_format(tmp_path, epinion2):
    from spex_amd.datasets import materialise_epinion2
    import lg_parser
    import utility1.dataloader as dl
    root = materialise_epinion2(str(tmp_path))
    ld = dl.Loader(lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", root]))
    assert (ld.n_user, ld.m_item, len(ld.rec_train_data)) == (3185, 12407, 209304)
    assert len(ld.testRatings) == 3185 and all(len(v) == 99 for v in ld.testNegatives.values())
    assert [ld.testRatings[u][0] for u in ld.testRatings] == epinion2["test_pos"].tolist()
    rowptr, col, val = ld.build_adjacency()
    assert len(col) == 418608


def test_path_attention_layer_closed_form_equals_reference_loops():
    """The closed form the HIP path-attention kernel implements (oracle/trust_oracle.py) against the reference's own
    loop structure (layers.py:15-71), on CPU tensors."""
    from oracle.trust_oracle import path_attention
    from utility2.layers import GraphAttentionLayer
    torch.manual_seed(0)
    H, B, L, U = 8, 5, 6, 20
    emb = torch.randn(U + 1, H)
    seq_l = torch.tensor([2, 6, 3, 5, 4])
    seq = torch.full((B, L), U)
    for p in range(B):
        seq[p, :seq_l[p]] = torch.randperm(U)[:seq_l[p]]

    def loops(layer, emb, seq, seq_l, concat):
        out = torch.ones(B, L, H)
        a = layer.a.detach()
        for p in range(B):
            l = int(seq_l[p])
            for i in range(l - 1):
                if concat:
                    x = emb[seq[p][i]] + float(l - i)
                    y = emb[seq[p][i + 1]] + float(l - i - 1)
                else:
                    x, y = seq[p][i], seq[p][i + 1]
                h = torch.stack([torch.cat([x, x]), torch.cat([x, y])])
                att = torch.softmax((h @ a).squeeze(1), dim=0)
                out[p][i] = att[0] * x + att[1] * y
            for i in range(l - 1, L):
                out[p][i] = emb[seq[p][i]] if concat else seq[p][i]
        return out

    la = GraphAttentionLayer(H, concat=True)
    assert torch.allclose(path_attention(emb, seq, seq_l, la.a.detach(), True), loops(la, emb, seq, seq_l, True), atol=1e-6)
    lb = GraphAttentionLayer(H, concat=False)
    x3 = torch.randn(B, L, H)
    assert torch.allclose(path_attention(None, x3, seq_l, lb.a.detach(), False), loops(lb, emb, x3, seq_l, False), atol=1e-6)
    with pytest.raises(RuntimeError):
        la(emb, seq, seq_l)                      # the product layer has no CPU path


def test_trust_data_batches_like_the_reference(golden):
    from utility2.utils import Data
    g = golden("trust_tiny")
    lens = g["train_mask"].sum(1)
    paths = [r[:l].tolist() for r, l in zip(g["train_inputs"], lens)]
    d = Data((paths, g["train_targets"].tolist()), 50)
    assert np.array_equal(d.inputs, g["train_inputs"]) and np.array_equal(d.mask, g["train_mask"]) and d.len_max == 6
    sl = d.generate_batch(16)
    assert [len(s) for s in sl] == [16, 16, 8] and np.array_equal(np.concatenate(sl), np.arange(40))
    i, m, t = d.get_slice(sl[2])
    assert np.array_equal(i, g["train_inputs"][32:]) and np.array_equal(t, g["train_targets"][32:])


def test_epoch_order_is_the_dataloaders_own(tiny_loader):
    """spex_amd.trainer.dataloader_epoch_order replays the installed torch's DataLoader(shuffle=True) index order from
    the global RNG, epoch after epoch (so the on-device epoch loop trains on the batches the reference driver would)."""
    import utility1.dataloader as dl
    from torch.utils.data import DataLoader
    from spex_amd.trainer import dataloader_epoch_order
    ld, _ = tiny_loader
    td = dl.LightTrainData(ld.rec_train_data, ld.m_item, ld.train_mat)
    np.random.seed(3)
    td.ng_sample()
    loader = DataLoader(td, batch_size=256, shuffle=True)
    torch.manual_seed(2020)
    want = [torch.cat([torch.stack([u, i, l]) for u, i, l in loader], dim=1) for _ in range(2)]   # two epochs
    torch.manual_seed(2020)
    for epoch in range(2):
        order = dataloader_epoch_order(len(td)).numpy()
        got = np.stack([td.users_fill[order], td.items_fill[order], td.labels_fill_np[order]])
        assert np.array_equal(got, want[epoch].numpy())


def test_reference_stream_dropout_mask_is_the_references_draw(golden):
    """The "reference" edge-dropout stream (dropin LightGCN.dropout_stream / trainer.edge_dropout_mask) replays model.py:50:
    after the reference's set-up — set_seed(2020), both nn.Embedding constructions + xavier_uniform_ (model.py:32-35), the
    shuffled DataLoader's two seed draws (main_rec.py:30) — the first `torch.rand(nnz) + keep_prob` gives the keep mask the
    REFERENCE drew at step 0 of `main_rec.py --dropout 1 --keepprob 0.3` on Epinion2 (G12-dropout golden: sha-256, kept count,
    first 4 096 bits).  CPU only: this is the host half of the validation mode; the GPU tests replay the whole run."""
    import utility1.utils as utils
    from torch import nn
    from spex_amd.trainer import dataloader_epoch_order
    g = golden("lightgcn_epinion2_dropout")
    utils.set_seed(int(g["seed"]))
    eu, ei = nn.Embedding(3186, 64), nn.Embedding(12407, 64)
    nn.init.xavier_uniform_(eu.weight, gain=1)
    nn.init.xavier_uniform_(ei.weight, gain=1)
    dataloader_epoch_order(6 * 209304)
    from spex_amd.trainer import reference_keep_mask
    keep = reference_keep_mask([int(g["nnz"])], float(g["keepprob"]), "cpu").numpy().astype(bool)     # the product's own replay (host branch)
    assert int(keep.sum()) == int(g["mask0_kept"])
    assert np.array_equal(keep[:4096], g["mask0_head"])
    assert sha(keep.astype(np.uint8)) == str(g["mask0_sha"])


# ---------------------------------------------------------------------------------------------- NGCF host modules (config 4)
def _write_ngcf_files(root, name, pairs, test_pos, test_neg):
    rec = os.path.join(root, name, "rec")
    os.makedirs(rec, exist_ok=True)
    with open(os.path.join(rec, "train.txt"), "w") as f:
        for u in np.unique(pairs[:, 0]):
            f.write(str(u) + "".join(" %d" % i for i in pairs[pairs[:, 0] == u, 1]) + "\n")
    with open(os.path.join(rec, "test.txt"), "w") as f:
        for u, p in enumerate(test_pos):
            f.write("%d %d\n" % (u, p))
        f.write("\n")                                  # a malformed line is skipped, as the reference's try/except does
    with open(os.path.join(rec, "negative.txt"), "w") as f:
        for u, n in enumerate(test_neg):
            f.write(str(u) + "".join(" %d" % i for i in n) + "\n")
    return os.path.join(root, name)


def test_ngcf_parser_defaults_match_reference_flags():
    from spex_amd.dropin.ngcf.ngcf_parser import parse_args
    a = parse_args([])
    assert (a.dataset, a.embed_size, a.layer_size, a.batch_size, a.lr, a.mess_dropout, a.Ks, a.test_flag, a.adj_type, a.epoch,
            a.regs, a.data_path) == ("epinion2", 64, "[64]", 256, 0.001, "[0.1]", "[10,20,50]", "part", "norm", 50, "[1e-5]",
                                     "../Data/")
    b = parse_args(["--dataset", "twitter", "--layer_size", "[64,64]", "--nonhybrid"])
    assert b.dataset == "twitter" and b.layer_size == "[64,64]" and b.nonhybrid


def test_ngcf_data_object_matches_the_reference(golden, tmp_path):
    """utility.load_data.Data against the golden minted from NGCF_SPEX/code/utility/load_data.py: sizes, the three
    adjacency matrices bit for bit (sha-256 of rowptr / col / val), and the negative-sampling stream of one epoch for the
    same random.seed (the reference's train_sample under a serial pool)."""
    import random
    from spex_amd.dropin.ngcf.utility.load_data import Data
    g = golden("ngcf_small_epochs")
    path = _write_ngcf_files(str(tmp_path), "small", g["train_pairs"], g["test_pos"], g["test_neg"])
    d = Data(path, 256)
    assert (d.n_users, d.n_items, d.n_train, d.n_test) == (int(g["n_users"]), int(g["n_items"]), int(g["n_train"]), 300)
    assert d.exist_users == list(range(300)) and d.test_set[7] == [int(g["test_pos"][7])]
    assert d.neg_item[3] == g["test_neg"][3].tolist() and d.R.shape == (300, 200) and d.R.nnz == d.n_train
    for name, m in zip(("plain", "norm", "mean"), d.get_adj_mat()):
        assert [sha(m.indptr.astype(np.int32)), sha(m.indices.astype(np.int32)), sha(m.data.astype(np.float32))] == g[name + "_sha"].tolist()
    random.seed(int(g["seed"]))
    u, v, r = d.sample_epoch()
    assert len(u) == int(g["sample_len"]) == 6 * sum(len(d.train_items[k]) for k in range(256))    # whole 256-user blocks only
    assert [sha(u), sha(v), sha(r)] == g["sample_sha"].tolist()
    assert np.array_equal(np.stack([u[:4096], v[:4096], r[:4096].astype(np.int64)]), g["sample_head"])
    # per user: 5 x |pos| DISTINCT negatives outside the user's items, then the positives in file order
    k = len(d.train_items[0])
    assert len(set(v[:5 * k].tolist())) == 5 * k and not set(v[:5 * k].tolist()) & set(d.train_items[0])
    assert v[5 * k:6 * k].tolist() == d.train_items[0] and r[:6 * k].tolist() == [0.0] * (5 * k) + [1.0] * k
    loader = d.load_train_data()
    assert loader.batch_size == 256 and len(loader.dataset) == len(u)


def test_ngcf_eval_ranking_semantics(golden, tmp_path, monkeypatch):
    """utility.batch_test.test_torch with the scoring launch replaced by exact host dot products (the GPU path is
    covered in tests/test_gpu_ngcf.py): the reference's per-user ranking — candidates = negatives then held-out items,
    ties broken by candidate order, recall over all held-out items — on hand-made tables."""
    from spex_amd.dropin.ngcf.utility import batch_test
    from spex_amd.dropin.ngcf.utility.load_data import Data
    g = golden("ngcf_small_epochs")
    path = _write_ngcf_files(str(tmp_path), "small", g["train_pairs"], g["test_pos"], g["test_neg"])
    d = Data(path, 256)
    batch_test.use_data(d)
    rng = np.random.default_rng(3)
    ua = torch.from_numpy(rng.integers(-3, 4, size=(d.n_users, 8)).astype(np.float32))     # small integers: exact, many ties
    ia = torch.from_numpy(rng.integers(-3, 4, size=(d.n_items, 8)).astype(np.float32))
    monkeypatch.setattr(batch_test, "_scores", lambda a, b, us, its: (a.numpy()[us] * b.numpy()[its]).sum(1))
    got = batch_test.test_torch(ua, ia, list(d.test_set.keys()))
    import heapq
    want = {"recall": np.zeros(3), "ndcg": np.zeros(3)}
    rate = ua.numpy() @ ia.numpy().T
    for u in d.test_set:                                  # the reference's test_one_user, restated (batch_test.py:91-116)
        items = d.neg_item[u] + d.test_set[u]
        score = {i: rate[u, i] for i in items}
        top = heapq.nlargest(50, score, key=score.get)
        r = [1 if i in d.test_set[u] else 0 for i in top]
        for j, k in enumerate((10, 20, 50)):
            want["recall"][j] += sum(r[:k]) / len(d.test_set[u]) / len(d.test_set)
            dcg = sum(x / np.log2(p + 2) for p, x in enumerate(r[:k]))
            idcg = sum(x / np.log2(p + 2) for p, x in enumerate(sorted(r, reverse=True)[:k]))
            want["ndcg"][j] += (dcg / idcg if idcg else 0.0) / len(d.test_set)
    assert np.abs(got["recall"] - want["recall"]).max() <= 1e-12 and np.abs(got["ndcg"] - want["ndcg"]).max() <= 1e-12
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        monkeypatch.undo()
        batch_test.test_torch(ua, ia, list(d.test_set.keys()))


def test_lightgcn_adjacency_sums_repeated_pairs_like_useritemnet():
    """A train file that repeats a pair: the reference's UserItemNet = csr_matrix((ones, (u, i))) sums the duplicates
    (dataloader.py:110), so the entry weighs 2 and both degrees count it twice."""
    import scipy.sparse as sp
    from spex_amd.graph import lightgcn_norm_adj
    u, i, nu, mi = np.array([0, 0, 1, 2, 2, 2]), np.array([1, 1, 0, 2, 2, 0]), 3, 3
    R = sp.csr_matrix((np.ones(len(u)), (u, i)), shape=(nu + 1, mi)).tolil()
    adj = sp.dok_matrix((nu + 1 + mi, nu + 1 + mi), dtype=np.float32).tolil()
    adj[:nu + 1, nu + 1:] = R
    adj[nu + 1:, :nu + 1] = R.T
    adj = adj.todok()
    with np.errstate(divide="ignore"):
        dinv = np.power(np.array(adj.sum(axis=1)), -0.5).flatten()
    dinv[np.isinf(dinv)] = 0.0
    want = sp.diags(dinv).dot(adj).dot(sp.diags(dinv)).tocsr()
    want.sort_indices()
    rowptr, col, val = lightgcn_norm_adj(u, i, nu, mi)
    assert np.array_equal(rowptr, want.indptr) and np.array_equal(col, want.indices)
    assert np.array_equal(val, want.data.astype(np.float32))


def test_message_dropout_mask_is_the_oracles_philox(oracle):
    """Known-answer test of the philox4x32-10 restatement (Random123's published test vectors), so that the mask the
    goldens were minted with is pinned independently of the kernels."""
    k = oracle.philox4x32_10
    assert [int(x[0]) for x in k([0], [0], [0], [0], 0, 0)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(x[0]) for x in k([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)] == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(x[0]) for x in k([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)] == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    m = oracle.message_keep_mask(500, 64, 0.1, 2020, 3, 0)
    assert m.shape == (500, 64) and abs(m.mean() - 0.9) < 0.01
    assert not np.array_equal(m, oracle.message_keep_mask(500, 64, 0.1, 2020, 4, 0))


def test_threaded_packer_lays_the_matrix_out_like_the_single_thread_planner():
    """spex_graph_create's host packer fills the chunk arrays on up to 16 threads (SPEX_BUILD_THREADS) from ranges a planning pass
    hands out; the layout must not depend on the thread count (VERDICT r2 asked whether the 2^24-node launch's 18.0 -> 19.1 ms
    between rounds 1 and 2 — "the same kernel, only the packer threaded" — was a different table: it is not).  Host only
    (spex_graph_pack_digest: FNV-1a of every array that would be uploaded): a 2.4 M-entry heavy-tailed graph — above the 2^20
    entries at which the packer goes parallel, with hub rows > 1 024 entries and empty rows — packed with 1, 3, 8 and 16 threads."""
    import ctypes
    from spex_amd import _lib
    from spex_amd.datasets import synthetic_interactions
    from spex_amd.graph import lightgcn_norm_adj
    u, i = synthetic_interactions(20000, 60000, 1500000, seed=3, sigma=1.3)
    rowptr, col, val = lightgcn_norm_adj(u.numpy(), i.numpy(), 20000, 60000)
    assert np.diff(rowptr).max() > 1024
    deg = np.diff(rowptr)
    assert len(col) > (1 << 20) and (deg == 0).any()
    rowptr, col, val = (np.ascontiguousarray(a) for a in (rowptr.astype(np.int32), col.astype(np.int32), val.astype(np.float32)))
    lib = _lib.load()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    digests = {}
    old = os.environ.get("SPEX_BUILD_THREADS")
    try:
        for n_thr in (1, 3, 8, 16):
            os.environ["SPEX_BUILD_THREADS"] = str(n_thr)
            d = (ctypes.c_uint64 * 8)()
            rc = lib.spex_graph_pack_digest(p(rowptr), p(col), p(val), len(rowptr) - 1, len(rowptr) - 1, len(col), d)
            assert rc == 0, lib.spex_last_error()
            digests[n_thr] = tuple(d)
    finally:
        if old is None:
            os.environ.pop("SPEX_BUILD_THREADS", None)
        else:
            os.environ["SPEX_BUILD_THREADS"] = old
    assert len(set(digests.values())) == 1, digests
    assert all(x != 0 for x in digests[1][:4])


def test_reference_keep_mask_draws_fold_by_fold_like_the_reference():
    """reference_keep_mask over several handles (--A_split: model.py:57-64 draws one mask per fold, in fold order) consumes the global
    generator exactly like the per-fold `torch.rand(len(values)) + keep_prob` calls it replaces."""
    from spex_amd.trainer import reference_keep_mask
    torch.manual_seed(77)
    want = torch.cat([(torch.rand(n) + 0.3).int().bool() for n in (1000, 7, 333)]).numpy()
    after = torch.rand(1).item()
    torch.manual_seed(77)
    got = reference_keep_mask([1000, 7, 333], 0.3, "cpu").numpy().astype(bool)
    assert np.array_equal(got, want) and torch.rand(1).item() == after


def test_message_dropout_replay_draws_what_nn_dropout_draws():
    """The NGCF validation mode (NGCF.dropout_stream / NGCFStepper.dropout_stream = "reference") rests on one fact about torch: on the
    CPU nn.Dropout(p) in training mode is x * (empty_like(x).bernoulli_(1 - p) / (1 - p)) — ONE bernoulli_ draw of x's shape from the
    global generator (main_rec.py:81 on the [N, 64] layer output).  Replaying `buf.bernoulli_(1 - p)` on a buffer of the same shape
    must give nn.Dropout's keep pattern and leave the generator where nn.Dropout leaves it; a torch that draws differently fails here,
    on the CPU, before the GPU golden does."""
    n, p = 1237, 0.1
    x = torch.rand(n, 64) + 0.5                          # no zeros: the output's zeros are exactly the dropped entries
    drop = torch.nn.Dropout(p)
    drop.train()
    torch.manual_seed(2020)
    out = [drop(x), drop(x)]                             # two layers / two steps in a row
    after = torch.rand(1).item()
    torch.manual_seed(2020)
    buf = torch.empty(n, 64)
    for o in out:
        keep = buf.bernoulli_(1.0 - p) != 0
        assert torch.equal(keep, o != 0)
        scale = torch.ones(()) / (1.0 - p)                      # at::dropout: noise.div_(1 - p), then x * noise — the kernels' `scale`
        assert torch.equal(o[keep], (x * scale)[keep])
    assert torch.rand(1).item() == after


def _hub_table(rowptr, col, val, n_cols):
    import ctypes
    from spex_amd import _lib
    lib = _lib.load()
    rowptr, col, val = (np.ascontiguousarray(a) for a in (rowptr.astype(np.int32), col.astype(np.int32), val.astype(np.float32)))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    n = ctypes.c_int64(0)
    args = (p(rowptr), p(col), p(val), len(rowptr) - 1, n_cols, len(col))
    assert lib.spex_graph_pack_hub_table(*args, None, 0, ctypes.byref(n)) == 0, lib.spex_last_error()
    out = np.zeros(n.value, np.int32)
    assert lib.spex_graph_pack_hub_table(*args, p(out), n.value, ctypes.byref(n)) == 0, lib.spex_last_error()
    n_pos, n_hubs = int(out[0]), int(out[1])
    return out[2:2 + 7 * n_pos].reshape(n_pos, 7), out[2 + 7 * n_pos:].reshape(n_hubs, 2)


@pytest.mark.parametrize("shape", ["mixed", "hubs-only", "no-hub"])
def test_packer_lays_hubs_out_in_groups_the_kernel_can_fold(shape):
    """What the d == 64 launch assumes when it folds the rows beyond 1 024 entries by itself (spmm.hip, HubFold), checked on the HOST
    (spex_graph_pack_hub_table): every hub starts a 16-task workgroup; its 64-entry segments follow in order, 16 to a group, the last
    group shorter and its workgroup topped up with ordinary (or null) tasks that carry the barrier bit; the first task of a group is
    its leader and knows the group's size; a hub's scratch rows are consecutive, one per group, and the fold table agrees."""
    rng = np.random.default_rng(21)
    if shape == "mixed":
        n_rows, n_cols = 500, 30000
        deg = rng.integers(0, 70, n_rows)
        for r, d_ in ((0, 1025), (1, 1024), (7, 2048), (8, 2049), (250, 20000), (499, 5000)):
            deg[r] = d_
    elif shape == "hubs-only":
        n_rows, n_cols = 3, 9000
        deg = np.array([3000, 1030, 8999])
    else:
        n_rows, n_cols = 200, 5000
        deg = rng.integers(0, 1025, n_rows)
    rowptr = np.zeros(n_rows + 1, np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = np.concatenate([np.sort(rng.choice(n_cols, d_, replace=False)) for d_ in deg] + [np.zeros(0, np.int64)])
    val = rng.random(len(col)).astype(np.float32)
    tasks, fold = _hub_table(rowptr, col, val, n_cols)
    hubs = [r for r in range(n_rows) if deg[r] > 1024]
    assert len(fold) == len(hubs)
    if not hubs:
        assert len(tasks) == 0
        return
    pos, slot = 0, 0
    for h, r in enumerate(hubs):
        n_seg = -(-int(deg[r]) // 64)
        n_grp = -(-n_seg // 16)
        assert pos % 16 == 0                                            # a hub starts a workgroup
        assert tuple(fold[h]) == (slot, n_grp)
        left_entries = int(deg[r])
        for gi in range(n_grp):
            size = min(16, n_seg - 16 * gi)
            for w in range(size):
                chunks, row, word, leader, waves, part, hub = (int(x) for x in tasks[pos])
                assert row == r and (word & 3) == 2 and (word & 4)      # a hub segment, barrier bit on
                assert chunks == -(-min(64, left_entries) // 16)        # 64-entry segments in row order (the last one ragged)
                left_entries -= min(64, left_entries)
                assert (leader, waves, part, hub) == (1 if w == 0 else 0, size, slot, h)
                pos += 1
            while pos % 16:                                             # the rest of the group's workgroup: ordinary / null tasks
                if pos < len(tasks):
                    chunks, row, word, leader, waves, part, hub = (int(x) for x in tasks[pos])
                    assert (word & 3) != 2 and (word & 4) and (leader, waves) == (0, 0)
                pos += 1
            slot += 1
        assert left_entries == 0
    assert pos >= len(tasks) > pos - 16


@pytest.mark.parametrize("edge_ids", [False, True])
def test_partition_push_structure_is_the_block_of_the_transpose_read_by_column(edge_ids):
    """Host logic of the partitioned fast path (spex_amd/dist.py: PartitionedLightGCN.push_graph, no GPU): the structure the exchange-free
    first backward product walks is the transpose of the rank's block of A^T — row p (a position of the padded global layout) holds
    exactly the entries A[g(p), c] for the columns c the rank owns, with LOCAL column indices, the entries' values and, under edge
    dropout (edge_ids=True), the index of the entry of A each came from (so that A, A^T and the push drop the same edges:
    utility1/model.py:46-64).  Checked entry by entry on a NON-symmetric matrix with an uneven 3-way partition."""
    import types
    from spex_amd.dist import PartitionedLightGCN
    from spex_amd.graph import csr_transpose
    rng = np.random.default_rng(4)
    n = 57
    dense = (rng.random((n, n)) < 0.15) * rng.uniform(0.1, 1.0, (n, n))
    dense[5, :] = rng.uniform(0.1, 1.0, n)                                   # a full row and a full column
    dense[:, 9] = rng.uniform(0.1, 1.0, n)
    dense = dense.astype(np.float32)
    rowptr = np.zeros(n + 1, np.int32)
    col, val = [], []
    for r in range(n):
        nz = np.flatnonzero(dense[r])
        col.extend(nz); val.extend(dense[r, nz]); rowptr[r + 1] = len(col)
    col, val = np.asarray(col, np.int32), np.asarray(val, np.float32)
    made = []

    def factory(rp, c, v, n_cols, edge_id=None):
        g = types.SimpleNamespace(host=(np.asarray(rp), np.asarray(c), np.asarray(v)), n_cols=n_cols, n_rows=len(rp) - 1, edge_id=edge_id,
                                  masks=[], set_edge_mask=lambda *a: g.masks.append(a))
        made.append(g)
        return g
    bounds = np.array([0, 11, 40, n])
    world, rank = 3, 1
    t_csr = csr_transpose(rowptr, col, val, n)
    P = PartitionedLightGCN(rowptr, col, val, 20, 3, 64, rank, world, factory, "cpu", t_csr=t_csr, bounds=bounds, edge_ids=edge_ids)
    if edge_ids:
        P.set_edge_mask(2, None, 0.3, 77)                                    # set BEFORE the push structure exists: re-applied at creation
    push = P.push_graph()
    assert push is P.push_graph() and push.n_cols == P.n_local == 29 and push.n_rows == P.part.n_padded
    prp, pc, pv = push.host
    pos_of = P.part.to_padded(np.arange(n))                                  # global row -> padded position
    seen = 0
    entry_of = {(int(r), int(c)): k for r in range(n) for k, c in zip(range(rowptr[r], rowptr[r + 1]), col[rowptr[r]:rowptr[r + 1]])}
    for g_row in range(n):
        p = int(pos_of[g_row])
        cols_local = pc[prp[p]:prp[p + 1]]
        want = [c for c in col[rowptr[g_row]:rowptr[g_row + 1]] if bounds[rank] <= c < bounds[rank + 1]]
        assert [int(c) + int(bounds[rank]) for c in cols_local] == [int(c) for c in want], g_row     # ascending, local indices
        for k, c in zip(range(prp[p], prp[p + 1]), cols_local):
            gc = int(c) + int(bounds[rank])
            assert pv[k] == dense[g_row, gc]
            if edge_ids:
                assert int(push.edge_id[k]) == entry_of[(g_row, gc)]
            seen += 1
    padded_rows = set(int(x) for x in pos_of)
    for p in range(P.part.n_padded):                                         # the slots' padding positions hold nothing
        if p not in padded_rows:
            assert prp[p + 1] == prp[p]
    assert seen == len(pc) == int((dense[:, bounds[rank]:bounds[rank + 1]] != 0).sum())
    if edge_ids:
        assert push.masks == [(2, None, 0.3, 77)] and P.graph.masks[-1] == (2, None, 0.3, 77) and P.graph_t.masks[-1] == (2, None, 0.3, 77)
        P.set_edge_mask(0)
        assert push.masks[-1][0] == 0
    else:
        assert push.edge_id is None


def test_train_epochs_draws_the_generators_like_a_sequential_loop():
    """Host logic of trainer.train_epochs (no GPU: a recording stand-in for the stepper): the next epoch's negatives (the train data's
    ng_sample: NumPy's global generator) and shuffle (dataloader_epoch_order: torch's global generator) are drawn on a second thread
    while the current epoch is being issued — and must come out exactly as a sequential `ng_sample(); train_epoch()` loop draws them
    (main_rec.py:25-31): the same batches in the same order, epoch after epoch."""
    from spex_amd.trainer import train_epoch, train_epochs

    class Data:
        def __init__(self):
            self.n = 1000
        def ng_sample(self):
            self.users_fill = np.random.randint(0, 50, self.n).astype(np.int64)
            self.items_fill = np.random.randint(0, 70, self.n).astype(np.int64)
            self.labels_fill_np = (np.random.random(self.n) < 0.2).astype(np.float32)
        def __len__(self):
            return self.n

    class Recorder:
        def __init__(self):
            self.E0 = torch.zeros(1)
            self.graph = self.graph_t = None
            self.seen = []
        def _one_call_ok(self, *a):
            return False                                     # (no native epoch on the CPU: the Python loop around step_bce)
        def step_bce(self, users, items, labels, loss_acc=None, batch_rows_only=False):
            self.seen.append((users.numpy().copy(), items.numpy().copy(), labels.numpy().copy()))
            loss_acc += float(len(users))
    runs = []
    for overlapped in (True, False):
        np.random.seed(5); torch.manual_seed(5)
        rec, td = Recorder(), Data()
        if overlapped:
            totals = train_epochs(rec, td, 3, batch_size=256)
        else:
            totals = [float(train_epoch(rec, td, batch_size=256)) for _ in range(3)]
        runs.append((rec.seen, totals))
    (a, ta), (b, tb) = runs
    assert len(a) == len(b) == 3 * 4 and ta == tb
    for x, y in zip(a, b):
        assert all(np.array_equal(p, q) for p, q in zip(x, y))
    assert not np.array_equal(a[0][0], a[4][0])                                       # (the epochs do differ)


def test_ngcf_blocked_sampler_replay_is_random_sample_draw_for_draw(epinion2):
    """NGCF's epoch sampler (NGCF_SPEX/code/utility/load_data.py:13-23,176-195: per user `random.sample(all_items - positives,
    5 |positives|)`) replayed in blocks on the generator's own output stream (dropin/ngcf/utility/load_data.py: _sample_epoch_blocked:
    the selection-set form as one vectorised pass per user, the pool-swap form of the heavy users draw by draw) against the
    draw-by-draw loop: on Epinion2 (3 072 users in whole blocks of 256, 63 of them on the pool-swap form) the same 1.2 M samples, and
    `random` left in the same state — so the NEXT epoch is the same too; plus the tiny cases (k <= 5, a user with one item)."""
    import random
    from collections import defaultdict
    import spex_amd.dropin.ngcf.utility.load_data as ld
    tr = epinion2["train"]
    ti = defaultdict(list)
    for u, i in tr:
        ti[int(u)].append(int(i))
    ti = dict(ti)
    n_items = int(tr[:, 1].max()) + 1
    ai = set(range(n_items))
    users = list(ti.keys())
    users = users[: len(users) // 256 * 256]
    assert ld._blocked_replay_applies(ai, ti)
    random.seed(2020)
    vs = []
    for u in users:
        vs.extend(ld.train_sample(u, ti, ai)[1])
    state_slow = random.getstate()
    random.seed(2020)
    fu, fv, fr = ld._sample_epoch_blocked(users, ti, ai)
    assert np.array_equal(np.asarray(vs), fv) and random.getstate() == state_slow
    assert len(fu) == len(fv) == len(fr) == 6 * sum(len(ti[u]) for u in users) and fr.sum() == sum(len(ti[u]) for u in users)
    small = {0: [1], 1: [0, 2, 3], 2: list(range(0, 40, 2)), 3: [5]}
    random.seed(3)
    want = []
    for u in small:
        want.extend(ld.train_sample(u, small, set(range(120)))[1])
    st = random.getstate()
    random.seed(3)
    got = ld._sample_epoch_blocked(list(small), small, set(range(120)))
    assert np.array_equal(np.asarray(want), got[1]) and random.getstate() == st
    # item ids with GAPS, the set grown by updates as Data grows it (items seen in the training file), a positive listed twice
    gappy = {0: [3, 40, 7], 1: [90, 3, 3, 55], 2: [7], 3: list(range(100, 160, 3))}
    seen = set()
    for its in gappy.values():
        seen.update(its)
    seen.update(range(200, 420, 2))
    assert ld._blocked_replay_applies(seen, gappy)
    random.seed(11)
    want = []
    for u in gappy:
        want.extend(ld.train_sample(u, gappy, seen)[1])
    st = random.getstate()
    random.seed(11)
    got = ld._sample_epoch_blocked(list(gappy), gappy, seen)
    assert np.array_equal(np.asarray(want), got[1]) and random.getstate() == st
    assert not ld._blocked_replay_applies(list(range(9)), small) and not ld._blocked_replay_applies({"a", "b"}, {0: ["a"]})   # (the loop)
    assert not ld._blocked_replay_applies({1, 2, 3, 10 ** 6}, {0: [1]})          # an id beyond the set's table: it does not iterate ascending
