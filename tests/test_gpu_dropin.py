"""The reference's Python surface on the GPU: the call sequence of LightGCN_SPEX/code/main_rec.py (Loader ->
LightTrainData -> LightGCN(args, dataset).to(device) -> Adam -> forward/backward/step -> test()) against the golden
vectors the reference produced for the same seeds."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def data_root(tmp_path_factory, golden):
    from spex_amd.datasets import materialise_rating_files, materialise_epinion2
    root = str(tmp_path_factory.mktemp("data"))
    g = golden("lightgcn_tiny")
    materialise_rating_files(root, "tiny", g["train_pairs"], g["test_users"], g["test_pos"], g["test_neg"])
    return materialise_epinion2(root)


def build(ds, data_root, extra=()):
    import lg_parser
    import utility1.dataloader as dataloader
    import utility1.model as model
    import utility1.utils as utils
    args = lg_parser.parse_args_r(["--dataset", ds, "--data_path", data_root, *extra])
    utils.set_seed(args.seed)                                   # main_rec.py:15
    dataset = dataloader.Loader(args)                           # :18
    net = model.LightGCN(args, dataset).to(DEV)                 # :22
    return args, dataset, net


def test_tiny_same_seed_same_tables_same_outputs(data_root, golden):
    g = golden("lightgcn_tiny")
    args, dataset, net = build("tiny", data_root)
    E0 = torch.cat([net.embedding_user.weight, net.embedding_item.weight]).detach().cpu().numpy()
    assert np.array_equal(E0, g["E0"])                          # same initialiser stream as model.py:32-35
    assert net.embedding_user.weight.data_ptr() + net.embedding_user.weight.numel() * 4 == net.embedding_item.weight.data_ptr()
    net.eval()
    with torch.no_grad():
        users, items = net.computer()
    assert users.shape == (51, 64) and items.shape == (60, 64)
    assert np.array_equal(torch.cat([users, items]).cpu().numpy(), g["light_out"])


def test_training_loop_and_eval_match_reference(data_root, golden):
    """main_rec.py:25-60 with the golden batches in place of the shuffled DataLoader."""
    sys.argv = ["main_rec.py"]
    from utility1.batch_test import test
    for ds in ("tiny", "epinion2"):
        g = golden(f"lightgcn_{ds}")
        args, dataset, net = build(ds, data_root)
        if ds == "epinion2":                                    # fixture names E0 by seed
            from spex_amd.datasets import epinion2_tables
            uw, iw = epinion2_tables(3186, 12407)
            with torch.no_grad():
                net.embedding_user.weight.copy_(torch.from_numpy(uw)); net.embedding_item.weight.copy_(torch.from_numpy(iw))
        opt = torch.optim.Adam(net.parameters(), lr=args.lr)    # :23
        net.train()
        for s in range(5):
            opt.zero_grad()
            u, i, y = (torch.from_numpy(g[k][s]) for k in ("batch_users", "batch_items", "batch_labels"))
            if s == 0:
                gamma = net(users=u.to(DEV), items=i.to(DEV), labels=y.to(DEV), flag=1)
                assert rel_err(gamma.cpu().numpy(), g["g3_gamma"]) <= 2e-6
            loss = net(users=u.to(DEV), items=i.to(DEV), labels=y.to(DEV), flag=0)   # :34
            loss.backward()                                                          # :35
            assert abs(loss.item() - float(g["g4_losses"][s])) <= 2e-6
            if s == 0:
                grad = torch.cat([net.embedding_user.weight.grad, net.embedding_item.weight.grad]).cpu().numpy()
                got = grad if ds == "tiny" else grad[g["sample_rows"]]
                assert rel_err(got, g["g3_grad"] if ds == "tiny" else g["g3_grad_rows"]) <= 1e-5
            opt.step()                                                               # :37
            if s + 1 in (1, 2, 5):
                W = torch.cat([net.embedding_user.weight, net.embedding_item.weight]).detach().cpu().numpy()
                got = W if ds == "tiny" else W[g["sample_rows"]]
                assert rel_err(got, g[f"g4_w_step{s + 1}"] if ds == "tiny" else g[f"g4_w_step{s + 1}_rows"]) <= 5e-6
        net.eval()
        with torch.no_grad():                                                        # :50
            ret = test(net, dataset.testRatings, dataset.testNegatives)
            # the acceptance gate: HR@K / NDCG@K within 1e-4 of the reference
            assert np.abs(ret["recall"] - g["g5_recall"]).max() <= 1e-4
            assert np.abs(ret["ndcg"] - g["g5_ndcg"]).max() <= 1e-4
            for k, uu in enumerate(g["g5_users"][:16]):
                its = dataset.testNegatives[int(uu)] + dataset.testRatings[int(uu)]
                sc = net(torch.full((len(its),), int(uu)).long(), torch.tensor(its).long(), None, flag=1)
                assert rel_err(sc.cpu().numpy(), g["g5_scores"][k]) <= 1e-4


def test_eval_cache_and_invalidation(data_root):
    args, dataset, net = build("tiny", data_root)
    net.eval()
    with torch.no_grad():
        a = net._light_out()
        b = net._light_out()
        assert a.data_ptr() == b.data_ptr()                      # one propagation for the whole eval loop
        with torch.no_grad():
            net.embedding_item.weight.add_(1.0)                  # any in-place parameter update invalidates it
        c = net._light_out()
        assert c.data_ptr() != a.data_ptr() or not torch.equal(a, c)
    net.train()
    x = net._light_out()
    assert x.requires_grad


def test_dropout_training_path(data_root, golden, oracle):
    g = golden("lightgcn_tiny")
    args, dataset, net = build("tiny", data_root, ["--dropout", "1", "--keepprob", "0.6"])
    keep = oracle.dropout_keep_mask(g["g9_rand"], 0.6)
    net.train()
    net.set_edge_mask(torch.from_numpy(keep))
    users, items = net.computer()
    assert rel_err(torch.cat([users, items]).detach().cpu().numpy(), g["g9_light_out"]) <= 1e-6
    # gradient through the masked, non-symmetric operator: compare with dense autograd on the same masked matrix
    rowptr, col, val = dataset.build_adjacency()
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    A = torch.zeros(n, n, dtype=torch.float64)
    A[rows[keep], col[keep]] = torch.from_numpy((val[keep] / np.float32(0.6)).astype(np.float64))
    E0 = torch.from_numpy(g["E0"]).double().requires_grad_(True)
    cur, acc = E0, E0
    for _ in range(3):
        cur = A @ cur
        acc = acc + cur
    w = torch.randn(n, 64, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    ((acc / 4) * w).sum().backward()
    (torch.cat([users, items]) * w.float().to(DEV)).sum().backward()
    got = torch.cat([net.embedding_user.weight.grad, net.embedding_item.weight.grad]).cpu().numpy()
    assert rel_err(got, E0.grad.numpy()) <= 1e-5
    # sampled masks: a fresh mask per step, finite loss, eval ignores dropout
    net.set_edge_mask(None)
    l1 = net(torch.from_numpy(g["batch_users"][0]), torch.from_numpy(g["batch_items"][0]), torch.from_numpy(g["batch_labels"][0]), flag=0)
    l2 = net(torch.from_numpy(g["batch_users"][0]), torch.from_numpy(g["batch_items"][0]), torch.from_numpy(g["batch_labels"][0]), flag=0)
    assert torch.isfinite(l1) and torch.isfinite(l2) and l1.item() != l2.item()
    net.eval()
    with torch.no_grad():
        u2, i2 = net.computer()
    assert np.array_equal(torch.cat([u2, i2]).cpu().numpy(), g["light_out"])


def test_a_split_folds_match_unsplit(data_root, golden):
    g = golden("lightgcn_tiny")
    args, dataset, net = build("tiny", data_root, ["--A_split", "1", "--a_fold", "7"])
    assert isinstance(net.Graph, list) and len(net.Graph) == 7
    net.eval()
    with torch.no_grad():
        users, items = net.computer()
    assert rel_err(torch.cat([users, items]).cpu().numpy(), g["light_out"]) <= 1e-6


@pytest.mark.parametrize("dropout", [False, True])
def test_a_split_training_equals_unsplit(data_root, golden, oracle, dropout):
    """--A_split trains (model.py:84-89 runs the folds under autograd too): forward on the row blocks of A, backward on
    the row blocks of A^T.  A row's sum does not depend on which block holds it, so loss, propagated table and
    gradients are BIT-IDENTICAL to the unsplit model — also under edge dropout, where the folds, the unsplit graph and
    the transposed blocks all look the same entries up in one mask (edge ids = positions in the unsplit matrix).
    (Checked bit for bit on the propagated table; the loss and the gradient start from the scoring kernel's float atomics,
    whose order varies from launch to launch, so they are compared to 1e-6.)"""
    g = golden("lightgcn_tiny")
    extra = ["--dropout", "1", "--keepprob", "0.6"] if dropout else []
    bu, bi, bl = (torch.from_numpy(g[k][0]) for k in ("batch_users", "batch_items", "batch_labels"))
    results = []
    for split in ([], ["--A_split", "1", "--a_fold", "7"]):
        args, dataset, net = build("tiny", data_root, extra + split)
        if dropout:
            net.set_edge_mask(torch.from_numpy(oracle.dropout_keep_mask(g["g9_rand"], 0.6)))
        net.train()
        loss = net(bu, bi, bl, flag=0)
        loss.backward()
        grad = torch.cat([net.embedding_user.weight.grad, net.embedding_item.weight.grad]).cpu().numpy()
        table = torch.cat(net.computer()).detach().cpu().numpy()
        results.append((loss.item(), grad, table))
    (l0, g0, t0), (l1, g1, t1) = results
    assert np.array_equal(t0, t1) and abs(l0 - l1) <= 1e-6 and rel_err(g1, g0) <= 1e-6
    if not dropout:
        assert abs(l0 - float(g["g3_loss"])) <= 2e-6 and rel_err(g1, g["g3_grad"]) <= 1e-5      # and both equal the reference
    # an optimiser step through the folds moves the parameters
    args, dataset, net = build("tiny", data_root, ["--A_split", "1", "--a_fold", "7"])
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    before = net.embedding_item.weight.detach().clone()
    net.train()
    net(bu, bi, bl, flag=0).backward()
    opt.step()
    assert not torch.equal(before, net.embedding_item.weight.detach())


def test_bpr_loss_extension_against_torch_fp32(data_root, golden):
    """bpr_loss() has no reference counterpart (parity unpinned): check value and gradient against plain torch ops
    on the same device (dense adjacency, fp64)."""
    g = golden("lightgcn_tiny")
    args, dataset, net = build("tiny", data_root)
    rng = np.random.default_rng(0)
    u, p, n_ = (torch.from_numpy(rng.integers(0, hi, 128)) for hi in (50, 60, 60))
    net.train()
    loss, reg = net.bpr_loss(u, p, n_)
    (loss + 1e-4 * reg).backward()
    got = torch.cat([net.embedding_user.weight.grad, net.embedding_item.weight.grad]).cpu().numpy()
    A = net.Graph.to_torch_sparse().to_dense().double()
    E0 = torch.from_numpy(g["E0"]).double().requires_grad_(True)
    cur, acc = E0, E0
    for _ in range(3):
        cur = A @ cur
        acc = acc + cur
    out = acc / 4
    U, I = out[:51], out[51:]
    x = (U[u] * I[n_]).sum(1) - (U[u] * I[p]).sum(1)
    ref_loss = torch.nn.functional.softplus(x).mean()
    ref_reg = 0.5 * (E0[:51][u].pow(2).sum() + E0[51:][p].pow(2).sum() + E0[51:][n_].pow(2).sum()) / 128
    (ref_loss + 1e-4 * ref_reg).backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-6 and abs(reg.item() - ref_reg.item()) <= 1e-6
    assert rel_err(got, E0.grad.numpy()) <= 1e-5


def test_unmodified_style_driver_runs_through_the_launcher(data_root, tmp_path):
    """A driver written against the reference's import names and call sequence, run with `python -m spex_amd.dropin`
    exactly as INTEGRATION.md tells a user to run the reference's own main_rec.py."""
    script = os.path.join(REPO, "tests", "drivers", "rec_driver.py")
    out = subprocess.run([sys.executable, "-m", "spex_amd.dropin", script, "--dataset", "tiny", "--data_path", data_root,
                          "--epochs", "2"], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("Rec:")]
    assert len(lines) >= 2 and "recall=" in lines[-1]
    # ... and the run IS the reference's (G12 on `tiny`, minted from main_rec.py's own loop with the same seed): the printed loss
    # sum and the printed HR / NDCG of every epoch
    import re
    e12 = np.load(os.path.join(REPO, "tests", "golden", "lightgcn_tiny_epochs.npz"))
    losses = [l for l in out.stdout.splitlines() if re.fullmatch(r"\d+,\d+\.\d+", l)]
    assert len(losses) == 2
    for ep in range(2):
        assert abs(float(losses[ep].split(",")[1]) - e12["losses"][ep]) <= 2e-5 * e12["losses"][ep] + 2e-5
        nums = [float(x) for x in re.findall(r"-?\d+\.\d+", lines[ep].split(":", 2)[2])]
        assert np.abs(np.array(nums[:3]) - e12["recall"][ep]).max() <= 1.5e-4 and np.abs(np.array(nums[3:6]) - e12["ndcg"][ep]).max() <= 1.5e-4


def test_dual_task_driver_runs_through_the_launcher(data_root, golden):
    """main_auto_expert_s.py's call sequence (rec loader + trust pickles + model_expert_s + uncertainty-weighted loss +
    rec_test / trust_test5) as a driver run with `python -m spex_amd.dropin`, two epochs on the tiny graph."""
    import pickle
    g = golden("trust_tiny")
    tdir = os.path.join(data_root, "tiny", "trust")
    os.makedirs(tdir, exist_ok=True)
    lens = g["train_mask"].sum(1)
    with open(os.path.join(tdir, "train.txt"), "wb") as f:
        pickle.dump(([r[:l].tolist() for r, l in zip(g["train_inputs"], lens)], g["train_targets"].tolist()), f)
    tl = g["test_mask"].sum(1)
    with open(os.path.join(tdir, "test2.txt"), "wb") as f:
        pickle.dump(([r[:l].tolist() for r, l in zip(g["test_inputs"], tl)], g["test_targets"].tolist(),
                     g["test_negs"].tolist()), f)
    script = os.path.join(REPO, "tests", "drivers", "dual_driver.py")
    out = subprocess.run([sys.executable, "-m", "spex_amd.dropin", script, "--dataset", "tiny", "--data_path", data_root,
                          "--epochs", "2"], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = [l for l in out.stdout.splitlines() if l.startswith("Rec:")]
    trust = [l for l in out.stdout.splitlines() if l.startswith("Trust:")]
    losses = [l for l in out.stdout.splitlines() if l[:2] in ("0,", "1,")]
    assert len(rec) == 2 and len(trust) == 2 and len(losses) == 2
    l0, l1 = (float(l.split(",")[1]) + float(l.split(",")[2]) for l in losses)
    assert l1 < l0                                                      # both tasks' summed loss goes down
    # ... and the run IS the reference's (G13, minted from main_auto_expert_s.py's own loop with the same seed): the printed loss
    # sums of both tasks and the printed metrics of both tasks, epoch by epoch (5 / 4 printed decimals)
    import re
    e13 = golden("dual_tiny_epochs")
    for ep, line in enumerate(losses):
        _, a, b = line.split(",")
        assert abs(float(a) - e13["loss1"][ep]) <= 2e-5 * e13["loss1"][ep] + 2e-5, (ep, a, e13["loss1"][ep])
        assert abs(float(b) - e13["loss2"][ep]) <= 5e-5 * e13["loss2"][ep] + 2e-5, (ep, b, e13["loss2"][ep])
    for ep in range(2):
        nums = [float(x) for x in re.findall(r"-?\d+\.\d+", rec[ep].split(":", 2)[2])]
        assert np.abs(np.array(nums[:3]) - e13["rec_recall"][ep]).max() <= 1.5e-4 and np.abs(np.array(nums[3:6]) - e13["rec_ndcg"][ep]).max() <= 1.5e-4
        tnums = [float(x) for x in re.findall(r"-?\d+\.\d+", trust[ep].split(":", 2)[2])]
        assert np.abs(np.array(tnums[:6]) - np.asarray(e13["trust"][ep]).ravel()[:6]).max() <= 1.5e-4


def _dual_task_model(data_root):
    import lg_parser
    import utility1.dataloader as dataloader
    import utility1.model_expert_s as mex
    import utility1.utils as utils
    args = lg_parser.parse_args_r(["--dataset", "tiny", "--data_path", data_root])
    utils.set_seed(args.seed)                                          # main_auto_expert_s.py:22
    dataset = dataloader.Loader(args)
    return args, dataset, mex.LightGCN(args, dataset)


def test_expert_gate_model_matches_reference_scores(data_root, golden):
    """model_expert_s rec branch (flag=1) with the reference's gate weights injected: G8."""
    from utility1.batch_test import rec_test
    g = golden("lightgcn_tiny")
    args, dataset, net = _dual_task_model(data_root)
    net = net.to(DEV)
    with torch.no_grad():
        E0 = torch.from_numpy(g["E0"]).to(DEV)
        net.embedding_user.weight.copy_(E0[:51]); net.embedding_item.weight.copy_(E0[51:])
        net.att_exp1.copy_(torch.from_numpy(g["g8_att_exp1"])); net.att_exp2.copy_(torch.from_numpy(g["g8_att_exp2"]))
    net.eval()
    u, i = torch.from_numpy(g["batch_users"][0]), torch.from_numpy(g["batch_items"][0])
    with torch.no_grad():
        gamma = net(u, i, None, None, None, flag=1)                       # fused gate kernel
    assert rel_err(gamma.cpu().numpy(), g["g8_gamma"]) <= 1e-5
    net.train()
    gamma_t = net(u, i, None, None, None, flag=1)                         # autograd (torch-op) gate agrees
    assert rel_err(gamma_t.detach().cpu().numpy(), g["g8_gamma"]) <= 1e-5
    net.eval()
    with torch.no_grad():
        ret = rec_test(net, dataset.testRatings, dataset.testNegatives)  # dual-task eval entry (batch_test.py:43-56)
    assert 0.0 <= ret["recall"][0] <= 1.0


def test_dual_task_model_matches_reference(data_root, golden):
    """G11: main_auto_expert_s.py's model — same parameters for the same seed, both losses, their gradients, trust
    scores and trust_test5 metrics, vs the reference (whose trust head is Python loops over batch x path position)."""
    from utility2.utils import Data
    from utility2.batch_test_gnn import trust_test5
    g = golden("trust_tiny")
    args, dataset, net = _dual_task_model(data_root)
    sd = net.state_dict()
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g["state_" + k.replace(".", "__")]), f"{k} differs from the reference init"
    net = net.to(DEV)
    lens = g["train_mask"].sum(1)
    train = Data(([r[:l].tolist() for r, l in zip(g["train_inputs"], lens)], g["train_targets"].tolist()), 50)
    assert np.array_equal(train.inputs, g["train_inputs"]) and np.array_equal(train.mask, g["train_mask"])
    tl = g["test_mask"].sum(1)
    test = Data(([r[:l].tolist() for r, l in zip(g["test_inputs"], tl)], g["test_targets"].tolist(),
                 g["test_negs"].tolist()), 50, test=True)
    gl = golden("lightgcn_tiny")
    bu, bi, bl = (torch.from_numpy(gl[k][0]) for k in ("batch_users", "batch_items", "batch_labels"))
    net.train()
    loss1, loss2 = net(bu, bi, bl, g["slice_indices"], train, flag=0)            # main_auto_expert_s.py:73-75
    assert abs(loss1.item() - float(g["loss1"])) <= 2e-6 and abs(loss2.item() - float(g["loss2"])) <= 2e-5
    (loss1 + loss2).backward()
    for name, p in net.named_parameters():
        key = "grad_" + name.replace(".", "__")
        if key in g.files:
            assert p.grad is not None, name
            assert rel_err(p.grad.cpu().numpy(), g[key]) <= 5e-5, name
    net.eval()
    with torch.no_grad():
        scores, negs = net(None, None, None, np.arange(12), test, flag=2)
    assert rel_err(scores.cpu().numpy(), g["trust_scores"]) <= 1e-5
    assert np.array_equal(negs.cpu().numpy(), g["trust_negs"])
    net.batch_size = 5
    got = np.asarray(trust_test5(net, test))
    assert np.abs(got - g["trust_test5"]).max() <= 1e-9


@pytest.mark.parametrize("form", ["fused", "fused-1wg", "fused-7wg"])
@pytest.mark.parametrize("nonhybrid", [False, True])
def test_fused_trust_head_matches_the_layer_by_layer_path(data_root, nonhybrid, form, monkeypatch):
    """The trust head's training call (spex_trust_head_train_f32) in all of its forms — the fused kernel split over the library's
    choice of workgroups per path; one workgroup per path; seven (SPEX_TRUST_SPLIT forces) — against the same model evaluated layer
    by layer (attention kernels + torch ops + autograd): loss, every parameter's gradient and the user-table gradient; paths of
    every length 1..L (a full-width path has no padded position for the max-pool's zero), repeated users, repeated targets."""
    from spex_amd import ops
    if form.endswith("wg"):
        monkeypatch.setenv("SPEX_TRUST_SPLIT", form[6:-2])
    args, dataset, net = _dual_task_model(data_root)
    net = net.to(DEV)
    net.nonhybrid = nonhybrid
    rng = np.random.default_rng(5)
    n_u, L = dataset.n_users, 7
    lens = np.r_[np.arange(1, L + 1), rng.integers(1, L + 1, 30)]
    inputs = np.full((lens.size, L), n_u, dtype=np.int64)
    for r, l in enumerate(lens):
        inputs[r, :l] = rng.integers(0, n_u, l)
    inputs[8, :3] = inputs[8, 0]                                      # the same user at several positions
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    targets = rng.integers(0, n_u, lens.size)
    with torch.no_grad():                                             # sizeable values so that every branch matters
        net.embedding_user.weight.mul_(8.0)
    params = [p for p in net.parameters() if p.requires_grad]

    def run(fused):
        for p in params:
            p.grad = None
        if fused:
            loss = net.trust_loss(inputs, mask, targets)
        else:
            scores = net._trust_scores(inputs, mask)
            loss = net.loss_function(scores, torch.from_numpy(targets).to(DEV))
        (loss * 1.7).backward()
        return loss.item(), {n: (p.grad.clone() if p.grad is not None else None) for n, p in net.named_parameters()}

    assert net._trust_fused_ok(L)
    l_ref, g_ref = run(False)
    l_fus, g_fus = run(True)
    assert abs(l_ref - l_fus) <= 2e-6 * max(1.0, abs(l_ref))
    checked = 0
    for name, ref in g_ref.items():
        if ref is None or not ref.abs().max().item():
            continue
        assert g_fus[name] is not None, name
        assert rel_err(g_fus[name].cpu().numpy(), ref.cpu().numpy()) <= 2e-5, name
        checked += 1
    assert checked >= (9 if nonhybrid else 11)
    # forward-only form (flag 2's scores) through the fused readout
    with torch.no_grad():
        flat = torch.cat([t.reshape(-1) for t in net._trust_param_tensors()])
        a2 = ops.trust_head_forward(net.embedding_user.weight, flat, torch.from_numpy(inputs), torch.from_numpy(lens), len(net.in_att),
                                    not nonhybrid)
        scores = net._trust_scores(inputs, mask)
        assert rel_err((a2 @ net.embedding_user.weight[:-1].t()).cpu().numpy(), scores.cpu().numpy()) <= 2e-5


def test_ngcf_model_matches_reference(golden, epinion2):
    """spex_amd.ngcf.NGCF with the reference's weights: forward (fused inference path and autograd path), loss and
    gradients vs G7."""
    import argparse
    import scipy.sparse as sp
    from spex_amd.graph import ngcf_norm_adj
    from spex_amd.ngcf import NGCF
    from spex_amd.datasets import epinion2_tables
    args = argparse.Namespace(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")
    for ds in ("tiny", "epinion2"):
        g = golden(f"ngcf_{ds}")
        nu, ni = int(g["n_users"]), int(g["n_items"])
        if ds == "tiny":
            csr, uw, iw = (g["rowptr"], g["col"], g["val"]), g["user_w"], g["item_w"]
        else:
            tr = epinion2["train"]
            csr = ngcf_norm_adj(tr[:, 0], tr[:, 1], nu, ni)
            uw, iw = epinion2_tables(nu + 1, ni)
        adj = sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(nu + ni, nu + ni))
        m = NGCF({"n_users": nu, "n_items": ni, "norm_adj": adj}, DEV, args).to(DEV)
        with torch.no_grad():
            m.user_embedding.weight.copy_(torch.from_numpy(uw)); m.item_embedding.weight.copy_(torch.from_numpy(iw))
            m.GC_Linear_list[0].weight.copy_(torch.from_numpy(g["W_gc"])); m.GC_Linear_list[0].bias.copy_(torch.from_numpy(g["b_gc"]))
            m.Bi_Linear_list[0].weight.copy_(torch.from_numpy(g["W_bi"])); m.Bi_Linear_list[0].bias.copy_(torch.from_numpy(g["b_bi"]))
        m.eval()
        rows = g["sample_rows"]
        want = g["all_emb"] if ds == "tiny" else g["all_emb_rows"]
        with torch.no_grad():
            ua, ia = m(None, None, None, flag=1)                          # fused kernels
        out = torch.cat([ua, ia]).cpu().numpy()
        assert rel_err(out if ds == "tiny" else out[rows], want) <= 1e-5
        ua2, ia2 = m(None, None, None, flag=1)                            # autograd path (dropout off in eval)
        out2 = torch.cat([ua2, ia2]).detach().cpu().numpy()
        assert rel_err(out2 if ds == "tiny" else out2[rows], want) <= 1e-5
        loss = m(torch.from_numpy(g["batch_users"]), torch.from_numpy(g["batch_items"]),
                 torch.from_numpy(g["batch_labels"]), flag=0)
        loss.backward()
        assert abs(loss.item() - float(g["loss"])) <= 2e-6
        assert rel_err(m.GC_Linear_list[0].weight.grad.cpu().numpy(), g["grad_W_gc"]) <= 2e-5
        assert rel_err(m.Bi_Linear_list[0].weight.grad.cpu().numpy(), g["grad_W_bi"]) <= 2e-5
        assert rel_err(m.GC_Linear_list[0].bias.grad.cpu().numpy(), g["grad_b_gc"]) <= 2e-5
        gall = torch.cat([m.user_embedding.weight.grad, m.item_embedding.weight.grad]).cpu().numpy()
        gr = g["grad_sample_rows"]
        assert rel_err(gall if ds == "tiny" else gall[gr], g["grad_emb"] if ds == "tiny" else g["grad_emb_rows"]) <= 2e-5
        assert np.isclose(np.sqrt((gall.astype(np.float64) ** 2).sum()), g["grad_emb_fro"], rtol=1e-4)


def test_on_device_epoch_equals_the_driver_loop(data_root):
    """spex_amd.trainer.train_epoch (negatives by the reference's sampler, the DataLoader's own shuffle order, one
    LightGCNStepper.step_bce per batch, nothing on the host between steps) against the reference-shaped driver loop
    (DataLoader + drop-in model + torch Adam, main_rec.py:25-38) for the same NumPy / torch seeds: same batches, same
    arithmetic => same loss sum and same tables after two epochs."""
    from torch.utils.data import DataLoader
    import utility1.dataloader as dl
    from spex_amd.trainer import LightGCNStepper, train_epoch
    results = []
    for fast in (False, True):
        args, dataset, net = build("tiny", data_root)
        td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
        np.random.seed(11)
        torch.manual_seed(12)
        if fast:
            st = LightGCNStepper(net.Graph, net.flat_table(), net.num_users + 1, n_layers=net.n_layers, lr=args.lr)
            total = 0.0
            for _ in range(2):
                total += train_epoch(st, td).item()
                # trained in place: the module's parameters ARE the stepper's table, and every eval sees the current ones
                # (the kernels' raw-pointer writes bump the tensors' version counters, which keys the eval-mode cache)
                net.eval()
                with torch.no_grad():
                    a = torch.cat(net.computer()).cpu().numpy()
                assert np.array_equal(a, net.Graph.propagate(st.E0, net.n_layers).cpu().numpy())
            results.append((total, torch.cat([net.embedding_user.weight, net.embedding_item.weight]).detach().cpu().numpy()))
        else:
            loader = DataLoader(td, batch_size=256, shuffle=True)
            opt = torch.optim.Adam(net.parameters(), lr=args.lr)
            net.train()
            total = 0.0
            for _ in range(2):
                loader.dataset.ng_sample()
                for user, item, label in loader:
                    opt.zero_grad()
                    loss = net(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV), flag=0)
                    loss.backward()
                    total += loss.item()
                    opt.step()
            results.append((total, torch.cat([net.embedding_user.weight, net.embedding_item.weight]).detach().cpu().numpy()))
    (l0, w0), (l1, w1) = results
    assert abs(l0 - l1) <= 1e-4 * abs(l0)
    assert rel_err(w1, w0) <= 2e-5


@pytest.mark.parametrize("ds,n_epochs", [("tiny", 3), ("epinion2", 1)])
def test_whole_training_run_matches_the_reference(data_root, golden, ds, n_epochs):
    """G12: the reference's own training run (main_rec.py:15-37,50 — set_seed, Loader, ng_sample, shuffled DataLoader,
    LightGCN, torch Adam, test(); three epochs on `tiny` and one full epoch — 4 906 steps — on Epinion2, minted by
    oracle/gen_golden.py --stage epochs / epochs-epinion2 from the reference's modules on CPU, 25 min for the latter)
    replayed through the drop-in modules on the GPU: the same first batch (same negatives,
    same shuffle), the same per-epoch loss sums, the same HR / NDCG after every epoch, the same trained tables — and
    the same again through the on-device epoch loop (trainer.train_epoch)."""
    from torch.utils.data import DataLoader
    import utility1.dataloader as dl
    from utility1.batch_test import test
    from spex_amd.trainer import LightGCNStepper, train_epoch
    g = golden(f"lightgcn_{ds}_epochs")
    for fast in (False, True):
        args, dataset, net = build(ds, data_root)                     # includes utils.set_seed(args.seed)
        assert args.seed == int(g["seed"]) and args.lr == float(g["lr"])
        td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
        loader = DataLoader(td, batch_size=256, shuffle=True)
        opt = torch.optim.Adam(net.parameters(), lr=args.lr)
        st = LightGCNStepper(net.Graph, net.flat_table(), net.num_users + 1, n_layers=net.n_layers, lr=args.lr)
        for epoch in range(n_epochs):
            if fast:
                total = train_epoch(st, td).item()
            else:
                loader.dataset.ng_sample()
                net.train()
                total = 0.0
                for k, (user, item, label) in enumerate(loader):
                    if epoch == 0 and k == 0:
                        assert np.array_equal(torch.stack([user, item, label]).numpy(), g["first_batch"])
                    opt.zero_grad()
                    loss = net(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV), flag=0)
                    loss.backward()
                    total += loss.item()
                    opt.step()
            assert abs(total - g["losses"][epoch]) <= 2e-5 * g["losses"][epoch], (fast, epoch, total)
            net.eval()
            with torch.no_grad():
                ret = test(net, dataset.testRatings, dataset.testNegatives)
            assert np.abs(ret["recall"] - g["recall"][epoch]).max() <= 1e-4
            assert np.abs(ret["ndcg"] - g["ndcg"][epoch]).max() <= 1e-4
        uw, iw = net.embedding_user.weight.detach().cpu().numpy(), net.embedding_item.weight.detach().cpu().numpy()
        if ds != "tiny":                                       # the full-size fixture stores sampled rows + column sums
            for got, want in ((uw, g["user_w_colsum"]), (iw, g["item_w_colsum"])):      # sums over all rows, |.| ~ 500
                assert np.abs(got.astype(np.float64).sum(0) - want).max() <= 2e-5 * np.abs(want).max()
            uw, iw = uw[g["rows_u"]], iw[g["rows_i"]]
        assert rel_err(uw, g["user_w"]) <= 1e-4 and rel_err(iw, g["item_w"]) <= 1e-4


@pytest.mark.parametrize("ds", ["tiny", "epinion2"])
def test_reference_stream_dropout_run_matches_the_reference(data_root, golden, ds):
    """G12-dropout: the reference's recommended LightGCN configuration, `main_rec.py --dropout 1 --keepprob 0.3`
    (README.md:119-123), minted from the reference's modules on CPU (oracle/gen_golden.py --stage epochs-dropout[-epinion2]:
    three epochs on tiny, the first 300 steps + test() on Epinion2).  The reference draws every step's edge mask with
    `torch.rand(nnz)` on the CPU from the global generator (model.py:46-55); the drop-in's "reference" dropout stream draws the
    same numbers at the same point, so the run drops the same edges and must reproduce EVERY step's loss (2e-5), the loss sum,
    HR / NDCG (1e-4) and the trained tables — through the unchanged-driver loop and through trainer.train_epoch."""
    from torch.utils.data import DataLoader
    import utility1.dataloader as dl
    from utility1.batch_test import test
    from spex_amd.trainer import LightGCNStepper, train_epoch
    g = golden(f"lightgcn_{ds}_dropout")
    keep_prob, max_steps = float(g["keepprob"]), (None if int(g["max_steps"]) < 0 else int(g["max_steps"]))
    n_epochs = len(g["losses"])
    for fast in (False, True):
        args, dataset, net = build(ds, data_root, ["--dropout", "1", "--keepprob", str(keep_prob)])   # includes set_seed
        net.dropout_stream = "reference"
        assert int(net.Graph.nnz) == int(g["nnz"])
        td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
        loader = DataLoader(td, batch_size=256, shuffle=True)
        opt = torch.optim.Adam(net.parameters(), lr=args.lr)
        st = LightGCNStepper(net.Graph, net.flat_table(), net.num_users + 1, n_layers=net.n_layers, lr=args.lr,
                             graph_t=net._transposed())
        step_losses = []
        for epoch in range(n_epochs):
            if fast:
                total = train_epoch(st, td, edge_dropout=(keep_prob, "reference"), max_steps=max_steps, step_losses=step_losses).item()
            else:
                loader.dataset.ng_sample()
                net.train()
                total = 0.0
                for k, (user, item, label) in enumerate(loader):
                    if max_steps is not None and k == max_steps:
                        break
                    if epoch == 0 and k == 0:
                        assert np.array_equal(torch.stack([user, item, label]).numpy(), g["first_batch"])
                    opt.zero_grad()
                    loss = net(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV), flag=0)
                    loss.backward()
                    step_losses.append(loss.item())
                    total += loss.item()
                    opt.step()
            assert abs(total - g["losses"][epoch]) <= 2e-5 * g["losses"][epoch], (fast, epoch, total, g["losses"][epoch])
            net.eval()
            with torch.no_grad():
                ret = test(net, dataset.testRatings, dataset.testNegatives)
            assert np.abs(ret["recall"] - g["recall"][epoch]).max() <= 1e-4, (fast, ret["recall"], g["recall"][epoch])
            assert np.abs(ret["ndcg"] - g["ndcg"][epoch]).max() <= 1e-4
        assert len(step_losses) == len(g["step_losses"])
        dev = np.abs(np.asarray(step_losses) - g["step_losses"])
        assert dev.max() <= 2e-5, (fast, int(dev.argmax()), float(dev.max()))
        uw, iw = net.embedding_user.weight.detach().cpu().numpy(), net.embedding_item.weight.detach().cpu().numpy()
        if ds != "tiny":
            for got, want in ((uw, g["user_w_colsum"]), (iw, g["item_w_colsum"])):
                assert np.abs(got.astype(np.float64).sum(0) - want).max() <= 2e-5 * np.abs(want).max()
            uw, iw = uw[g["rows_u"]], iw[g["rows_i"]]
        assert rel_err(uw, g["user_w"]) <= 5e-5 and rel_err(iw, g["item_w"]) <= 5e-5


@pytest.mark.parametrize("dropout", [None, (0.3, "philox", 11)])
def test_native_epoch_loop_equals_the_python_loop(data_root, dropout):
    """trainer.train_epoch issues the whole epoch as ONE native call (spex_lightgcn_epoch_bce_f32: batch k = samples [kB, (k+1)B) of
    the shuffled, device-resident epoch through the one-call step — Train() of main_rec.py:30-37) where nothing has to happen on the
    host between two steps; asking for per-step losses keeps the Python loop around the one-call step.  Same batches, same masks
    (the sampled edge-dropout stream is keyed by (seed, step) in the kernel), same arithmetic: 300 steps + a ragged tail agree in
    the loss sum and the trained table to the float atomics' reordering."""
    import utility1.dataloader as dl
    from spex_amd.trainer import LightGCNStepper, train_epoch
    extra = [] if dropout is None else ["--dropout", "1", "--keepprob", str(dropout[0])]
    out = []
    for native in (True, False):
        args, dataset, net = build("epinion2", data_root, extra)
        td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
        st = LightGCNStepper(net.Graph, net.flat_table(), net.num_users + 1, n_layers=net.n_layers, lr=args.lr,
                             graph_t=net._transposed() if dropout is not None else None)
        per_step = None if native else []
        total = train_epoch(st, td, edge_dropout=dropout, max_steps=300, step_losses=per_step).item()
        # a short "epoch" with a ragged last batch: 3 full batches + 77 samples
        class _Tail:
            users_fill, items_fill, labels_fill_np = td.users_fill[:845], td.items_fill[:845], td.labels_fill_np[:845]
            def __len__(self):
                return 845
        total2 = train_epoch(st, _Tail(), resample=False, edge_dropout=dropout, step_losses=None if native else []).item()
        assert st.t == 304 and getattr(st.graph, "mask_mode", 0) == 0
        out.append((total, total2, st.E0.detach().cpu().numpy().copy()))
    (a1, a2, Ea), (b1, b2, Eb) = out
    assert abs(a1 - b1) <= 2e-6 * abs(b1) and abs(a2 - b2) <= 2e-6 * abs(b2), (a1, b1, a2, b2)
    assert rel_err(Ea, Eb) <= 2e-5


def test_train_epochs_prepares_the_next_epoch_beside_the_current_one(data_root):
    """trainer.train_epochs: the next epoch's negatives (LightTrainData.ng_sample, NumPy's global stream) and shuffle (the DataLoader's
    order, torch's global stream) are drawn on a second host thread while the native epoch call trains the current one — in the order a
    sequential loop draws them (main_rec.py:25-37), so two epochs of 150 steps equal two sequential train_epoch calls: loss sums and
    the trained table to the float atomics' reordering."""
    import utility1.dataloader as dl
    from spex_amd.trainer import LightGCNStepper, train_epoch, train_epochs

    class _Short:            # the first 150 batches of the real training data (a whole epoch is 4 906)
        def __init__(self, td):
            self.td = td
        def ng_sample(self):
            self.td.ng_sample()
        def __len__(self):
            return 150 * 256
        users_fill = property(lambda self: self.td.users_fill[: 2 * len(self)])
        items_fill = property(lambda self: self.td.items_fill[: 2 * len(self)])
        labels_fill_np = property(lambda self: self.td.labels_fill_np[: 2 * len(self)])
    out = []
    for overlapped in (True, False):
        args, dataset, net = build("epinion2", data_root)
        td = _Short(dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat))
        st = LightGCNStepper(net.Graph, net.flat_table(), net.num_users + 1, n_layers=net.n_layers, lr=args.lr)
        if overlapped:
            seen = []
            totals = train_epochs(st, td, 2, after_epoch=lambda ep, t: seen.append(ep))
            assert seen == [0, 1]
        else:
            totals = [train_epoch(st, td).item() for _ in range(2)]
        assert st.t == 300
        out.append((totals, st.E0.detach().cpu().numpy().copy()))
    (ta, Ea), (tb, Eb) = out
    assert np.abs(np.asarray(ta) - np.asarray(tb)).max() <= 2e-6 * max(tb), (ta, tb)
    assert rel_err(Ea, Eb) <= 2e-5


@pytest.mark.parametrize("L", [2, 4])
def test_two_and_four_layer_runs_match_the_reference(data_root, golden, L):
    """The reference at another depth (`main_rec.py --layer 2` / `--layer 4`, lg_parser.py:10; oracle/gen_golden.py --stage
    epochs-L{2,4}-epinion2: the first 120 steps on Epinion2 + test(), minted from the reference's modules on CPU).  Every other
    LightGCN golden is the default L = 3, whose one-call step takes the all-plain schedule; L = 2 takes the plain forward with the
    mean's share of the one backward product added by the Adam pass, L = 4 the running-sum forward and the add-form backward
    products (utility1/model.py:83-97 and its autograd).  EVERY step's loss (3e-6), the loss sum, HR / NDCG (1e-4) and the trained
    tables — through the unchanged-driver loop (module + autograd + torch Adam) and through trainer.train_epoch (one native call per
    step)."""
    from torch.utils.data import DataLoader
    import utility1.dataloader as dl
    from utility1.batch_test import test
    from spex_amd.trainer import LightGCNStepper, train_epoch
    g = golden(f"lightgcn_epinion2_L{L}")
    max_steps = int(g["max_steps"])
    assert int(g["n_layers"]) == L and len(g["step_losses"]) == max_steps
    for fast in (False, True):
        args, dataset, net = build("epinion2", data_root, ["--layer", str(L)])        # includes set_seed
        assert net.n_layers == L
        td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
        step_losses = []
        if fast:
            st = LightGCNStepper(net.Graph, net.flat_table(), net.num_users + 1, n_layers=L, lr=args.lr)
            total = train_epoch(st, td, max_steps=max_steps, step_losses=step_losses).item()
        else:
            loader = DataLoader(td, batch_size=256, shuffle=True)
            opt = torch.optim.Adam(net.parameters(), lr=args.lr)
            loader.dataset.ng_sample()
            net.train()
            total = 0.0
            for k, (user, item, label) in enumerate(loader):
                if k == max_steps:
                    break
                if k == 0:
                    assert np.array_equal(torch.stack([user, item, label]).numpy(), g["first_batch"])
                opt.zero_grad()
                loss = net(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV), flag=0)
                loss.backward()
                step_losses.append(loss.item())
                total += loss.item()
                opt.step()
        assert abs(total - g["losses"][0]) <= 2e-5 * g["losses"][0], (fast, total, g["losses"][0])
        dev = np.abs(np.asarray(step_losses) - g["step_losses"])
        # (120 steps from the initial tables: every loss is within 2e-4 of log 2 and the depths differ by ~1e-5 in it — hence 3e-6, not
        #  the 2e-5 of the longer goldens; the depth shows plainly in HR / NDCG, 0.24 vs 0.30, and in the trained rows below)
        assert len(step_losses) == max_steps and dev.max() <= 3e-6, (fast, int(dev.argmax()), float(dev.max()))
        net.eval()
        with torch.no_grad():
            ret = test(net, dataset.testRatings, dataset.testNegatives)
        assert np.abs(ret["recall"] - g["recall"][0]).max() <= 1e-4, (fast, ret["recall"], g["recall"][0])
        assert np.abs(ret["ndcg"] - g["ndcg"][0]).max() <= 1e-4
        uw, iw = net.embedding_user.weight.detach().cpu().numpy(), net.embedding_item.weight.detach().cpu().numpy()
        for got, want in ((uw, g["user_w_colsum"]), (iw, g["item_w_colsum"])):
            assert np.abs(got.astype(np.float64).sum(0) - want).max() <= 2e-5 * np.abs(want).max()
        assert rel_err(uw[g["rows_u"]], g["user_w"]) <= 5e-5 and rel_err(iw[g["rows_i"]], g["item_w"]) <= 5e-5


@pytest.mark.parametrize("mode", ["philox", "reference"])
def test_one_call_step_under_edge_dropout_equals_the_launch_by_launch_step(data_root, mode):
    """spex_lightgcn_step_bce_f32 with an edge-dropout mask on both handles (round 3: the batch kernel's last layer and push apply
    the handles' keep rule; the whole-graph launches always did) against LightGCNStepper's launch-by-launch step (masked dense
    propagation, scoring, masked all-pull backward, Adam) on Epinion2: six steps with a fresh mask per step — in-kernel Philox
    draw, and an uploaded keep mask (the reference-stream form) — same losses, same tables to rounding (the one-call step adds
    duplicate rows and the push with float atomics)."""
    import utility1.dataloader as dl
    from spex_amd.trainer import LightGCNStepper, edge_dropout_mask
    args, dataset, net = build("epinion2", data_root, ["--dropout", "1", "--keepprob", "0.3"])
    rng = np.random.default_rng(3)
    B, steps = 256, 6
    deg = np.diff(dataset.build_adjacency()[0])
    users = rng.integers(0, dataset.n_users, (steps, B)); items = rng.integers(0, dataset.m_items, (steps, B))
    users[:, 0] = int(np.argmax(deg[: dataset.n_users]))                     # a hub user and a hub item in every batch
    items[:, 0] = int(np.argmax(deg[dataset.n_users + 1:]))
    users[:, 5:8] = users[:, 0:1]
    labels = (rng.random((steps, B)) < 1 / 6).astype(np.float32)
    u_d, i_d, y_d = (torch.from_numpy(a).to(DEV) for a in (users, items, labels))
    E0 = net.flat_table().detach().clone()
    out = []
    for one_call in (True, False):
        torch.manual_seed(11)                                                # the "reference" stream draws from the global generator
        st = LightGCNStepper(net.Graph, E0.clone(), net.num_users + 1, n_layers=net.n_layers, lr=args.lr, graph_t=net._transposed())
        acc = torch.zeros(1, device=DEV)
        per_step = []
        for k in range(steps):
            mask = edge_dropout_mask(net.Graph, 0.3, mode, 5, k + 1)
            st.graph.set_edge_mask(*mask)
            st.graph_t.set_edge_mask(*mask)
            before = acc.item()
            st.step_bce(u_d[k], i_d[k], y_d[k], loss_acc=acc, batch_rows_only=one_call)
            per_step.append(acc.item() - before)
        st.graph.set_edge_mask(0)
        st.graph_t.set_edge_mask(0)
        out.append((np.asarray(per_step), st.E0.detach().cpu().numpy()))
    (l_a, E_a), (l_b, E_b) = out
    assert np.abs(l_a - l_b).max() <= 2e-5 * np.abs(l_b).max()
    assert rel_err(E_a, E_b) <= 2e-5
    assert np.abs(E_a - E0.cpu().numpy()).max() > 1e-4                        # (the steps did move the table)


def test_dropped_entries_contribute_nothing_even_beside_non_finite_rows(data_root, golden):
    """A dropped edge is GONE from the reference's matrix (model.py:52-54); in the kernel it keeps its slot and gathers a
    stand-in row, whose value is replaced by 0 before the fmaf — so a table row holding Inf cannot turn into 0 * Inf = NaN on
    rows whose surviving entries never reference it (round 2 multiplied)."""
    g = golden("lightgcn_tiny")
    args, dataset, net = build("tiny", data_root, ["--dropout", "1", "--keepprob", "0.5"])
    rowptr, col, val = dataset.build_adjacency()
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    bad = int(np.bincount(col, minlength=n).argmax())                 # the most referenced source row
    keep = np.ones(len(col), bool)
    keep[col == bad] = False                                          # every edge INTO it is dropped ...
    keep[np.random.default_rng(0).random(len(col)) < 0.3] = False     # ... and a random third of the others
    keep[col == bad] = False
    E0 = g["E0"].copy()
    E0[bad] = np.inf
    net.train()
    net.set_edge_mask(torch.from_numpy(keep))
    with torch.no_grad():
        net.embedding_user.weight.copy_(torch.from_numpy(E0[: net.num_users + 1]))
        net.embedding_item.weight.copy_(torch.from_numpy(E0[net.num_users + 1:]))
        one = net.Graph.spmm
        net.Graph.set_edge_mask(1, net._injected_mask, 0.5, 0)
        y = net.Graph.spmm(net.flat_table())
        net.Graph.set_edge_mask(0)
    got = y.cpu().numpy()
    A = np.zeros((n, n), np.float64)
    A[rows[keep], col[keep]] = val[keep].astype(np.float64) / 0.5
    Ez = E0.astype(np.float64).copy()
    Ez[bad] = 0.0                                                     # no surviving entry references the bad row
    want = A @ Ez
    assert np.isfinite(got).all()
    assert rel_err(got, want) <= 1e-6


def test_whole_dual_task_run_matches_the_reference(data_root, golden):
    """G13: the reference's dual-task training run (main_auto_expert_s.py:22-120 driven from the reference's modules on
    CPU: rec batches, per-batch path selection incl. random.sample, uncertainty-weighted loss, Adam, rec_test +
    trust_test5; two epochs on `tiny`) replayed through the drop-in modules on the GPU: the same number of paths per
    batch, the same loss sums of both tasks, the same learned task weights, the same metrics of both tasks."""
    import random
    from collections import defaultdict
    from torch.utils.data import DataLoader
    import utility1.dataloader as dl
    from utility1.batch_test import rec_test
    from utility2.batch_test_gnn import trust_test5
    from utility2.utils import Data
    g, g11 = golden("dual_tiny_epochs"), golden("trust_tiny")
    lens, tl = g11["train_mask"].sum(1), g11["test_mask"].sum(1)
    raw_train = ([r[:l].tolist() for r, l in zip(g11["train_inputs"], lens)], g11["train_targets"].tolist())
    raw_test = ([r[:l].tolist() for r, l in zip(g11["test_inputs"], tl)], g11["test_targets"].tolist(), g11["test_negs"].tolist())
    args, dataset, net = _dual_task_model(data_root)               # includes utils.set_seed (also seeds `random`)
    assert args.seed == int(g["seed"])
    loader = DataLoader(dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat), batch_size=256, shuffle=True)
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2 = Data(raw_train, dataset.n_users, shuffle=False)
    test2 = Data(raw_test, dataset.n_users, shuffle=False, test=True)
    cap = 3 * (len(raw_train[0]) // len(loader))
    net = net.to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)
    step = 0
    for epoch in range(2):
        loader.dataset.ng_sample()
        net.train()
        t1 = t2 = 0.0
        for user, item, label in loader:
            opt.zero_grad()
            chosen = []
            for u in set(user.numpy().tolist()):
                chosen.extend(by_user[u])
            if len(chosen) > cap:
                chosen = random.sample(chosen, cap)
            assert len(chosen) == int(g["n_paths"][step])
            step += 1
            l1, l2 = net(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV),
                         slice_indices=np.array(chosen, dtype=int), trust_data=train2, flag=0)
            w = net.task_weights
            (torch.exp(-2 * w[0]) * l1 + torch.exp(-2 * w[1]) * l2 + 2 * 6 * len(user) * w[0] + len(chosen) * w[1]).backward()
            t1 += l1.item()
            t2 += l2.item()
            opt.step()
        assert abs(t1 - g["loss1"][epoch]) <= 2e-5 * g["loss1"][epoch] and abs(t2 - g["loss2"][epoch]) <= 5e-5 * g["loss2"][epoch]
        assert np.abs(net.task_weights.detach().cpu().numpy() - g["task_weights"][epoch]).max() <= 1e-6
        net.eval()
        with torch.no_grad():
            ret = rec_test(net, dataset.testRatings, dataset.testNegatives)
            assert np.abs(ret["recall"] - g["rec_recall"][epoch]).max() <= 1e-4 and np.abs(ret["ndcg"] - g["rec_ndcg"][epoch]).max() <= 1e-4
            assert np.abs(np.asarray(trust_test5(net, test2)) - g["trust"][epoch]).max() <= 1e-4
    assert rel_err(net.embedding_user.weight.detach().cpu().numpy(), g["user_w"]) <= 5e-5
    assert rel_err(net.w.detach().cpu().numpy(), g["w"]) <= 5e-5


# ---------------------------------------------------------------------------------------------- config 5 at Epinion2 scale
def _epinion2_trust_raw(golden):
    """The reference-minted Epinion2 trust paths (Data_process/path/data_process_path.py on trust_with_timestamp.mat;
    oracle/gen_golden.py --stage paths) as the lists main_auto_expert_s.py unpickles."""
    t = golden("trust_epinion2_paths")
    tr = ([r[:l].tolist() for r, l in zip(t["train_paths"].astype(np.int64), t["train_len"])],
          t["train_targets"].astype(np.int64).tolist())
    te = ([r[:l].tolist() for r, l in zip(t["test_paths"].astype(np.int64), t["test_len"])],
          t["test_targets"].astype(np.int64).tolist(), t["test_negs"].astype(np.int64).tolist())
    return tr, te


def _dual_epinion2(data_root, extra=()):
    import lg_parser
    import utility1.dataloader as dataloader
    import utility1.model_expert_s as mex
    import utility1.utils as utils
    args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", data_root, *extra])
    utils.set_seed(args.seed)
    dataset = dataloader.Loader(args)
    return args, dataset, mex.LightGCN(args, dataset)


def test_dual_task_model_matches_reference_epinion2(data_root, golden):
    """G11 at Epinion2 scale (3 185 users, 12 407 items, 27 004 reference-minted trust paths): same parameters for the
    same seed, both losses for a 256-sample rec batch + 192 paths, every parameter's gradient, the trust scores at the
    test negatives and trust_test5 over 1 024 test paths."""
    from collections import defaultdict
    from utility2.batch_test_gnn import trust_test5
    from utility2.utils import Data
    from test_oracle_golden import sha
    g, lg = golden("trust_epinion2"), golden("lightgcn_epinion2")
    raw_train, raw_test = _epinion2_trust_raw(golden)
    args, dataset, net = _dual_epinion2(data_root)
    assert sha(net.embedding_user.weight.detach().numpy()) == str(g["user_w_sha"])
    assert sha(net.embedding_item.weight.detach().numpy()) == str(g["item_w_sha"])
    for k in g.files:
        if k.startswith("state_"):
            assert np.array_equal(net.state_dict()[k[6:].replace("__", ".")].numpy(), g[k]), k
    net = net.to(DEV)
    train, test = Data(raw_train, dataset.n_users, shuffle=False), Data(raw_test, dataset.n_users, shuffle=False, test=True)
    bu, bi, bl = (torch.from_numpy(lg[k][0]) for k in ("batch_users", "batch_items", "batch_labels"))
    net.train()
    loss1, loss2 = net(bu, bi, bl, g["slice_indices"], train, flag=0)
    assert abs(loss1.item() - float(g["loss1"])) <= 2e-6 and abs(loss2.item() - float(g["loss2"])) <= 2e-5
    (loss1 + loss2).backward()
    checked = 0
    for name, p in net.named_parameters():
        key = name.replace(".", "__")
        if "grad_" + key not in g.files:
            continue
        got = p.grad.cpu().numpy()
        if "gradrows_" + key in g.files:
            assert abs(np.sqrt((got.astype(np.float64) ** 2).sum()) - float(g["gradfro_" + key])) <= 5e-5 * float(g["gradfro_" + key]), name
            cs = g["gradcolsum_" + key]
            assert np.abs(got.astype(np.float64).sum(0) - cs).max() <= 5e-5 * max(np.abs(cs).max(), 1e-6), name
            got = got[g["gradrows_" + key]]
        assert rel_err(got, g["grad_" + key]) <= 5e-5, name
        checked += 1
    assert checked >= 12
    net.eval()
    with torch.no_grad():
        scores, negs = net(None, None, None, np.arange(64), test, flag=2)
        assert rel_err(torch.gather(scores, 1, negs).cpu().numpy(), g["trust_scores_at_negs"]) <= 2e-5
        assert np.abs(np.asarray(trust_test5(net, test)) - g["trust_test5"]).max() <= 1e-4


def test_dual_task_training_run_matches_the_reference_epinion2(data_root, golden):
    """G13 at Epinion2 scale: the first 600 dual-task steps of main_auto_expert_s.py's epoch 0 (rec batches of 256, the
    batch users' trust paths capped at 3 x 5 by random.sample, uncertainty-weighted loss, Adam over all ~40 parameters),
    then rec_test over 3 185 users and trust_test5 — per-step path counts, per-step losses of the first 16 steps, running
    loss sums every 100 steps, the learned task weights, both tasks' metrics, the trained tables."""
    import random
    from collections import defaultdict
    from torch.utils.data import DataLoader
    import utility1.dataloader as dl
    from utility1.batch_test import rec_test
    from utility2.batch_test_gnn import trust_test5
    from utility2.utils import Data
    g = golden("dual_epinion2_epochs")
    raw_train, raw_test = _epinion2_trust_raw(golden)
    args, dataset, net = _dual_epinion2(data_root)
    assert args.seed == int(g["seed"])
    loader = DataLoader(dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat), batch_size=256, shuffle=True)
    assert len(loader) == int(g["steps_per_epoch"])
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2, test2 = Data(raw_train, dataset.n_users, shuffle=False), Data(raw_test, dataset.n_users, shuffle=False, test=True)
    cap = 3 * (len(raw_train[0]) // len(loader))
    assert cap == 3 * int(g["trust_batch_size"])
    net = net.to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)
    loader.dataset.ng_sample()
    net.train()
    t1 = t2 = 0.0
    n_steps = int(g["n_steps"])
    # fixed summation orders through the autograd path too: the library's Functions (ops.set_deterministic: batch rows added in
    # slot order, the gate's parameter gradient in block order) and torch's own index backward (deterministic index_add_)
    from spex_amd import ops
    ops.set_deterministic(True)
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        _dual_driver_loop_600(g, net, opt, loader, by_user, cap, train2, test2, dataset, n_steps)
    finally:
        ops.set_deterministic(False)
        torch.use_deterministic_algorithms(False)


def _dual_driver_loop_600(g, net, opt, loader, by_user, cap, train2, test2, dataset, n_steps):
    import random
    from utility1.batch_test import rec_test
    from utility2.batch_test_gnn import trust_test5
    t1 = t2 = 0.0
    for step, (user, item, label) in enumerate(loader):
        if step == n_steps:
            break
        opt.zero_grad()
        chosen = []
        for u in set(user.numpy().tolist()):
            chosen.extend(by_user[u])
        if len(chosen) > cap:
            chosen = random.sample(chosen, cap)
        assert len(chosen) == int(g["n_paths"][step]), step
        l1, l2 = net(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV),
                     slice_indices=np.array(chosen, dtype=int), trust_data=train2, flag=0)
        w = net.task_weights
        (torch.exp(-2 * w[0]) * l1 + torch.exp(-2 * w[1]) * l2 + 2 * 6 * len(user) * w[0] + len(chosen) * w[1]).backward()
        a, b = l1.item(), l2.item()
        if step < len(g["loss1_first"]):
            assert abs(a - g["loss1_first"][step]) <= 2e-5 and abs(b - g["loss2_first"][step]) <= 1e-4 * g["loss2_first"][step], step
        t1 += a
        t2 += b
        opt.step()
        if (step + 1) % 100 == 0:
            c = (step + 1) // 100 - 1
            assert abs(t1 - g["loss1_cum"][c]) <= 5e-5 * g["loss1_cum"][c], (step, t1, g["loss1_cum"][c])
            assert abs(t2 - g["loss2_cum"][c]) <= 2e-4 * g["loss2_cum"][c], (step, t2, g["loss2_cum"][c])
    net.eval()
    with torch.no_grad():
        ret = rec_test(net, dataset.testRatings, dataset.testNegatives)
        tr5 = np.asarray(trust_test5(net, test2))
    uw, iw = net.embedding_user.weight.detach().cpu().numpy(), net.embedding_item.weight.detach().cpu().numpy()
    dev = dict(task_w=float(np.abs(net.task_weights.detach().cpu().numpy() - g["task_weights"]).max()),
               rec=float(max(np.abs(ret["recall"] - g["rec_recall"]).max(), np.abs(ret["ndcg"] - g["rec_ndcg"]).max())),
               trust=float(np.abs(tr5 - g["trust"]).max()), user=rel_err(uw[g["rows_u"]], g["user_w"]),
               item=rel_err(iw[g["rows_i"]], g["item_w"]), w=rel_err(net.w.detach().cpu().numpy(), g["w"]))
    print("dual-task driver loop (autograd path, fixed summation orders), 600 steps, deviation from the reference's run:", dev)
    assert dev["task_w"] <= 5e-6 and dev["rec"] <= 1e-4 and dev["trust"] <= 1e-4, dev
    assert max(dev["user"], dev["item"], dev["w"]) <= 1e-4, dev


def test_dual_task_one_call_step_matches_the_autograd_step(data_root, golden):
    """spex_dual_task_step_f32 (DualTaskStepper: 14 launches, one library call) against the same step taken through the
    drop-in model with autograd and torch.optim.Adam: three steps on the tiny graph with the G11 batch and paths — every
    parameter (tables, ~15 trust-head tensors, gate matrices, task weights) and both losses."""
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper
    g, gl = golden("trust_tiny"), golden("lightgcn_tiny")
    args, dataset, net = _dual_task_model(data_root)
    _, _, ref = _dual_task_model(data_root)                         # the same seed: the same initial parameters
    net, ref = net.to(DEV), ref.to(DEV)
    lens = g["train_mask"].sum(1)
    train = Data(([r[:l].tolist() for r, l in zip(g["train_inputs"], lens)], g["train_targets"].tolist()), 50)
    sl = np.asarray(g["slice_indices"])
    inputs, mask, targets = train.get_slice(sl)
    bu, bi, bl = (torch.from_numpy(gl[k][0]).to(DEV) for k in ("batch_users", "batch_items", "batch_labels"))
    opt = torch.optim.Adam(ref.parameters(), lr=args.lr)
    st = DualTaskStepper(net, path_capacity=len(sl), path_len=inputs.shape[1], lr=args.lr)
    seq, seq_l, tgt = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(DEV) for a in (inputs, mask.sum(1), targets))
    ref.train()
    want = []
    for step in range(3):
        opt.zero_grad()
        l1, l2 = ref(bu, bi, bl, sl, train, flag=0)
        w = ref.task_weights
        (torch.exp(-2 * w[0]) * l1 + torch.exp(-2 * w[1]) * l2 + 2 * 6 * bu.numel() * w[0] + len(sl) * w[1]).backward()
        opt.step()
        want.append((l1.item(), l2.item()))
        st.loss_acc.zero_()
        st.step(bu, bi, bl.float(), seq, seq_l, tgt)
        got = st.loss_acc.cpu().numpy()
        assert abs(got[0] - want[-1][0]) <= 3e-6 and abs(got[1] - want[-1][1]) <= 2e-5, (step, got, want[-1])
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        # Adam's first steps move every touched parameter by ~lr whatever the gradient's size: compare on that scale
        assert (p.detach() - q.detach()).abs().max().item() <= 0.02 * args.lr * 3, name
    assert st.t == 3
    # the module is still the model: evaluation sees the trained values
    net.eval(); ref.eval()
    with torch.no_grad():
        a = net(bu, bi, None, None, None, flag=1)
        b = ref(bu, bi, None, None, None, flag=1)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-4


@pytest.mark.parametrize("name,extra,deterministic", [("dual_epinion2_epochs", (), True), ("dual_epinion2_L2_epochs", ("--layer", "2"), True),
                                                    ("dual_epinion2_L2_epochs", ("--layer", "2"), False),
                                                    ("dual_epinion2_L4_epochs", ("--layer", "4"), True), ("dual_epinion2_L4_epochs", ("--layer", "4"), False)])
def test_dual_task_on_device_epoch_matches_the_reference_epinion2(data_root, golden, name, extra, deterministic):
    """G13 at Epinion2 scale through the on-device epoch loop (trainer.train_epoch_dual + DualTaskStepper): the same 600
    steps as the reference's run (same negatives, same shuffle, same random.sample path cuts) — running loss sums, the
    learned task weights, both tasks' metrics and the trained tables.
    dual_epinion2_L{2,4}_epochs: the first 100 steps of `main_auto_expert_s.py --layer 2` / `--layer 4` (oracle/gen_golden.py --stage
    epochs-dual-L{2,4}-epinion2) — every other dual-task golden is the default depth 3, whose one-call step runs the all-plain backward;
    at L = 2 the single backward product is plain and the Adam pass adds the mean's share, at L = 4 the forward keeps its running sum
    and the middle backward products take the add form (utility1/model_expert_s.py:95-126 and its autograd) — in the deterministic
    mode and on the fast path (float atomics: same gates)."""
    from collections import defaultdict
    import utility1.dataloader as dl
    from utility1.batch_test import rec_test
    from utility2.batch_test_gnn import trust_test5
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper, train_epoch_dual
    g = golden(name)
    raw_train, raw_test = _epinion2_trust_raw(golden)
    args, dataset, net = _dual_epinion2(data_root, extra)
    td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2, test2 = Data(raw_train, dataset.n_users, shuffle=False), Data(raw_test, dataset.n_users, shuffle=False, test=True)
    cap = 3 * int(g["trust_batch_size"])
    net = net.to(DEV)
    st = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr, deterministic=deterministic)
    n_steps = int(g["n_steps"])
    totals = train_epoch_dual(st, td, train2, by_user, cap, max_steps=n_steps).cpu().numpy()
    assert st.t == n_steps
    c = n_steps // 100 - 1
    net.eval()
    with torch.no_grad():
        ret = rec_test(net, dataset.testRatings, dataset.testNegatives)
        tr5 = np.asarray(trust_test5(net, test2))
    dev = dict(loss1=abs(totals[0] - g["loss1_cum"][c]) / g["loss1_cum"][c], loss2=abs(totals[1] - g["loss2_cum"][c]) / g["loss2_cum"][c],
               task_w=float(np.abs(net.task_weights.detach().cpu().numpy() - g["task_weights"]).max()),
               rec=float(max(np.abs(ret["recall"] - g["rec_recall"]).max(), np.abs(ret["ndcg"] - g["rec_ndcg"]).max())),
               trust=float(np.abs(tr5 - g["trust"]).max()))
    # The deterministic step repeats bit for bit (tests/test_gpu_deterministic.py), so these deviations from the reference's
    # 600-step run are FIXED numbers — measured on the MI355X: loss sums 8.6e-7 / 3.0e-7, task weights 5.4e-7, HR / NDCG of the
    # rec task 2.8e-7 (no user changes rank), of the trust task 1e-15.  (Two mints of the reference's own run agree to 1e-8
    # here: LightGCN has no scale-invariant direction for Adam to amplify noise along.)  Round 2's gates of 1e-3 / 2e-3 on the
    # metrics absorbed the float atomics' reordering and nothing else; they are back at the north-star gate.
    print("dual-task run (%s, deterministic=%s), %d steps, deviation from the reference's run:" % (name, deterministic, n_steps), dev)
    assert dev["loss1"] <= 2e-5 and dev["loss2"] <= 2e-5 and dev["task_w"] <= 5e-6, dev
    assert dev["rec"] <= 1e-4 and dev["trust"] <= 1e-4, dev
    uw, iw = net.embedding_user.weight.detach().cpu().numpy(), net.embedding_item.weight.detach().cpu().numpy()
    tab = dict(user=rel_err(uw[g["rows_u"]], g["user_w"]), item=rel_err(iw[g["rows_i"]], g["item_w"]), w=rel_err(net.w.detach().cpu().numpy(), g["w"]))
    print("trained tables vs the reference's after %d steps:" % n_steps, tab)
    assert max(tab.values()) <= 1e-4, tab


def test_native_dual_task_epoch_and_overlapped_preparation_equal_the_python_loop(data_root, golden):
    """trainer.train_epoch_dual issues the epoch as ONE native call (spex_dual_task_epoch_f32: Train() of main_auto_expert_s.py:60-91,
    batch after batch with the staged paths of each) where nothing has to happen on the host between two steps; asking for running
    sums keeps the Python loop around the one-call step; trainer.train_epochs_dual prepares the next epoch's negatives, shuffle and
    path cuts on a second thread meanwhile.  DETERMINISTIC step (a run is a pure function of its inputs): 2 x 200 steps the three ways
    end in bit-identical parameters and loss sums; and under the sampled edge dropout the native loop equals the Python loop."""
    from collections import defaultdict
    import utility1.dataloader as dl
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper, train_epoch_dual, train_epochs_dual
    raw_train, _ = _epinion2_trust_raw(golden)
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)

    class _Short:            # the first 200 batches of the real training data (a whole epoch is 4 906)
        def __init__(self, td):
            self.td = td
        def ng_sample(self):
            self.td.ng_sample()
        def __len__(self):
            return 200 * 256
        users_fill = property(lambda self: self.td.users_fill[: 2 * len(self)])
        items_fill = property(lambda self: self.td.items_fill[: 2 * len(self)])
        labels_fill_np = property(lambda self: self.td.labels_fill_np[: 2 * len(self)])
    out = {}
    for way in ("native", "python", "overlapped", "native-dropout", "python-dropout"):
        drop = ("--dropout", "1", "--keepprob", "0.3") if way.endswith("dropout") else ()
        args, dataset, net = _dual_epinion2(data_root, drop)
        td = _Short(dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat))
        train2 = Data(raw_train, dataset.n_users, shuffle=False)
        cap = 3 * (len(raw_train[0]) // 4906)
        net = net.to(DEV)
        st = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr, deterministic=True)
        ed = (0.3, "philox", 5) if drop else None
        if way == "overlapped":
            totals = train_epochs_dual(st, td, train2, by_user, cap, 2)
        else:
            kw = dict(cum_every=10 ** 9, cum_out=[]) if way.startswith("python") else {}
            totals = [train_epoch_dual(st, td, train2, by_user, cap, edge_dropout=ed, **kw).cpu().numpy() for _ in range(2)]
        assert st.t == 400 and getattr(net.Graph, "mask_mode", 0) == 0
        out[way] = (np.stack(totals), [p.detach().clone() for p in net.parameters()])
    for a, b in (("python", "native"), ("overlapped", "native"), ("python-dropout", "native-dropout")):
        assert np.array_equal(out[a][0], out[b][0]), (a, out[a][0], out[b][0])
        assert all(torch.equal(x, y) for x, y in zip(out[a][1], out[b][1])), a
    assert not np.array_equal(out["native"][0], out["native-dropout"][0])


def test_dual_task_run_under_edge_dropout_matches_the_reference_epinion2(data_root, golden):
    """G13-dropout: main_auto_expert_s.py under the reference's recommended `--dropout 1 --keepprob 0.3` (README.md:119-123) on
    Epinion2, 150 steps minted from the reference's modules (oracle/gen_golden.py --stage dual-dropout-epinion2).  The rec branch's
    edge mask is the reference's own per-step `torch.rand(nnz)` draw, replayed from the global CPU generator ("reference" stream) and
    applied by the one-call step (spex_dual_task_step_f32 with the mask on both handles: the whole-graph launches AND the fused
    batch kernel's last layer / push drop the same edges).  Every step's two losses, the learned task weights, both tasks' HR /
    NDCG and the trained tables — through trainer.train_epoch_dual in the deterministic mode."""
    import hashlib
    from collections import defaultdict
    import lg_parser
    import utility1.dataloader as dl
    import utility1.model_expert_s as mex
    import utility1.utils as utils
    from utility1.batch_test import rec_test
    from utility2.batch_test_gnn import trust_test5
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper, train_epoch_dual
    g = golden("dual_epinion2_dropout")
    keep_prob, n_steps = float(g["keepprob"]), int(g["n_steps"])
    raw_train, raw_test = _epinion2_trust_raw(golden)
    args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", data_root, "--dropout", "1", "--keepprob", str(keep_prob)])
    utils.set_seed(args.seed)
    dataset = dl.Loader(args)
    net = mex.LightGCN(args, dataset)
    assert int(net.Graph.nnz) == int(g["nnz"])
    td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2, test2 = Data(raw_train, dataset.n_users, shuffle=False), Data(raw_test, dataset.n_users, shuffle=False, test=True)
    cap = 3 * int(g["trust_batch_size"])
    net = net.to(DEV)
    st = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr, deterministic=True)
    cum, n_paths = [], []
    train_epoch_dual(st, td, train2, by_user, cap, max_steps=n_steps, cum_every=1, cum_out=cum, n_paths_out=n_paths,
                     edge_dropout=(keep_prob, "reference"))
    assert st.t == n_steps and np.array_equal(np.asarray(n_paths), g["n_paths"])
    cum = torch.stack(cum).cpu().numpy().astype(np.float64)
    per_step = np.diff(np.concatenate([np.zeros((1, 2)), cum]), axis=0)
    d1, d2 = np.abs(per_step[:, 0] - g["step_loss1"]), np.abs(per_step[:, 1] - g["step_loss2"]) / g["step_loss2"]
    # (per-step values are differences of an fp32 running sum that reaches ~100 / ~900: 1e-5 / 1e-4 of resolution)
    assert d1.max() <= 5e-5, (int(d1.argmax()), float(d1.max()))
    assert d2.max() <= 2e-4, (int(d2.argmax()), float(d2.max()))
    assert abs(cum[-1, 0] - g["step_loss1"].sum()) <= 2e-5 * g["step_loss1"].sum()
    assert abs(cum[-1, 1] - g["step_loss2"].sum()) <= 2e-5 * g["step_loss2"].sum()
    assert np.abs(net.task_weights.detach().cpu().numpy() - g["task_weights"]).max() <= 5e-6
    net.eval()
    with torch.no_grad():
        ret = rec_test(net, dataset.testRatings, dataset.testNegatives)
        tr5 = np.asarray(trust_test5(net, test2))
    assert max(np.abs(ret["recall"] - g["rec_recall"]).max(), np.abs(ret["ndcg"] - g["rec_ndcg"]).max()) <= 1e-4
    assert np.abs(tr5 - g["trust"]).max() <= 1e-4
    uw, iw = net.embedding_user.weight.detach().cpu().numpy(), net.embedding_item.weight.detach().cpu().numpy()
    assert rel_err(uw[g["rows_u"]], g["user_w"]) <= 1e-4 and rel_err(iw[g["rows_i"]], g["item_w"]) <= 1e-4
    assert rel_err(net.w.detach().cpu().numpy(), g["w"]) <= 1e-4
    # the same run's first 40 steps through the UNCHANGED driver's loop (main_auto_expert_s.py:60-89 on the drop-in model: autograd,
    # torch.optim.Adam, the model drawing its mask from the reference stream inside forward)
    import random
    from torch.utils.data import DataLoader
    utils.set_seed(args.seed)
    dataset2 = dl.Loader(args)
    net2 = mex.LightGCN(args, dataset2).to(DEV)
    net2.dropout_stream = "reference"
    td2 = dl.LightTrainData(dataset2.rec_train_data, dataset2.m_item, dataset2.train_mat)
    loader = DataLoader(td2, batch_size=256, shuffle=True)
    opt = torch.optim.Adam(net2.parameters(), lr=args.lr)
    loader.dataset.ng_sample()
    net2.train()
    for step, (user, item, label) in enumerate(loader):
        if step == 40:
            break
        if step == 0:
            assert np.array_equal(torch.stack([user, item, label]).numpy(), g["first_batch"])
        opt.zero_grad()
        chosen = []
        for uu in set(user.numpy().tolist()):
            chosen.extend(by_user[uu])
        if len(chosen) > cap:
            chosen = random.sample(chosen, cap)
        assert len(chosen) == int(g["n_paths"][step])
        l1, l2 = net2(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV), slice_indices=np.array(chosen, dtype=int),
                      trust_data=train2, flag=0)
        w_ = net2.task_weights
        (torch.exp(-2 * w_[0]) * l1 + torch.exp(-2 * w_[1]) * l2 + 2 * 6 * len(user) * w_[0] + len(chosen) * w_[1]).backward()
        assert abs(l1.item() - g["step_loss1"][step]) <= 2e-5, (step, l1.item(), g["step_loss1"][step])
        assert abs(l2.item() - g["step_loss2"][step]) <= 1e-4 * g["step_loss2"][step], (step, l2.item(), g["step_loss2"][step])
        opt.step()


def test_dual_task_teacher_forced_checkpoint_epinion2(data_root, golden):
    """Teacher forcing at TRAINED weights for config 5: the reference's full dual-task parameter state after 600 steps of
    main_auto_expert_s.py on Epinion2 (oracle/gen_golden.py --stage epochs-dual-epinion2[-full] -> dual_epinion2_ckpt.npz: all
    ~20 parameters in fp32, step 600's rec batch and path selection, both losses, every gradient of the uncertainty-weighted
    loss, rec_test + trust_test5 at that state).  Loaded into the GPU model: one forward / backward -> loss1 <= 2e-5, loss2 <=
    1e-4 relative (a cross-entropy of ~5 over 3 185 logits), gradients <= 5e-5; one DualTaskStepper step in both accumulation
    modes -> the same losses; rec_test / trust_test5 -> HR / NDCG <= 1e-4 (the north-star gate, at trained weights)."""
    from utility1.batch_test import rec_test
    from utility2.batch_test_gnn import trust_test5
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper
    g = golden("dual_epinion2_ckpt")
    tag = "ckpt%d" % int(g["ckpt_step"])
    raw_train, raw_test = _epinion2_trust_raw(golden)
    args, dataset, net = _dual_epinion2(data_root)
    train2, test2 = Data(raw_train, dataset.n_users, shuffle=False), Data(raw_test, dataset.n_users, shuffle=False, test=True)
    net = net.to(DEV)

    def load_state():
        with torch.no_grad():
            for name, p in net.named_parameters():
                p.copy_(torch.from_numpy(g[f"{tag}_state_" + name.replace(".", "__")]).reshape(p.shape))
        net._cache = None
    load_state()
    user, item, label = (torch.from_numpy(g[f"{tag}_batch"][k]) for k in range(3))
    paths = g[f"{tag}_path_index"].astype(int)
    want1, want2 = float(g[f"{tag}_loss1"]), float(g[f"{tag}_loss2"])
    net.train()
    net.zero_grad()
    l1, l2 = net(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV), slice_indices=paths, trust_data=train2, flag=0)
    w = net.task_weights
    (torch.exp(-2 * w[0]) * l1 + torch.exp(-2 * w[1]) * l2 + 2 * 6 * len(user) * w[0] + len(paths) * w[1]).backward()
    assert abs(l1.item() - want1) <= 2e-5 and abs(l2.item() - want2) <= 1e-4 * want2, (l1.item(), want1, l2.item(), want2)
    checked = 0
    for name, p in net.named_parameters():
        key = f"{tag}_grad_" + name.replace(".", "__")
        if key not in g.files:
            continue
        got = p.grad.cpu().numpy()
        if key + "_rows" in g.files:
            fro = float(g[key + "_fro"])
            assert abs(np.sqrt((got.astype(np.float64) ** 2).sum()) - fro) <= 5e-5 * fro, name
            cs = g[key + "_colsum"]
            assert np.abs(got.astype(np.float64).sum(0) - cs).max() <= 5e-5 * max(np.abs(cs).max(), 1e-6), name
            got = got[g[key + "_rows"]]
        want = g[key].reshape(got.shape)
        # (att_t: the two columns of a two-way softmax's parameter gradient are exact negatives of each other; ours are, the
        #  reference's differ from each other by 1e-7 absolute — its own rounding — hence the absolute alternative)
        assert rel_err(got, want) <= 5e-5 or np.abs(got - want).max() <= 5e-7, (name, rel_err(got, want))
        checked += 1
    assert checked >= 14
    # ---- the one-call step at the same state (both accumulation modes): both losses
    inputs, mask, targets = train2.get_slice(paths)
    seq, seq_l, tgt = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(DEV) for a in (inputs, np.asarray(mask).sum(1), targets))
    for det in (False, True):
        load_state()
        st = DualTaskStepper(net, path_capacity=len(paths), path_len=train2.len_max, lr=args.lr, deterministic=det)
        st.step(user.to(DEV), item.to(DEV), label.float().to(DEV), seq, seq_l, tgt)
        got = st.loss_acc.cpu().numpy()
        assert abs(got[0] - want1) <= 2e-5 and abs(got[1] - want2) <= 1e-4 * want2, (det, got, want1, want2)
        del st
    # ---- both evaluations at the reference's trained weights
    load_state()
    net.eval()
    with torch.no_grad():
        ret = rec_test(net, dataset.testRatings, dataset.testNegatives)
        assert np.abs(np.concatenate([ret["recall"], ret["ndcg"]]) - g["metrics_rec"]).max() <= 1e-4
        assert np.abs(np.asarray(trust_test5(net, test2)) - g["metrics_trust"]).max() <= 1e-4


def test_dual_task_full_epoch_matches_the_reference_epinion2(data_root, golden):
    """G13, the WHOLE epoch: all 4 906 dual-task steps of main_auto_expert_s.py's epoch 0 on Epinion2 (oracle/gen_golden.py --stage
    epochs-dual-epinion2-full, ~20 min of reference CPU time) through trainer.train_epoch_dual with the deterministic step: the
    per-step path counts, both running loss sums every 100 steps, the learned task weights, rec_test + trust_test5 after the epoch,
    the trained parameters — and, teacher-forced, both evaluations at the REFERENCE's end-of-epoch state (1e-4)."""
    from collections import defaultdict
    import utility1.dataloader as dl
    from utility1.batch_test import rec_test
    from utility2.batch_test_gnn import trust_test5
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper, train_epoch_dual
    path = os.path.join(REPO, "tests", "golden", "dual_epinion2_full_epoch.npz")
    if not os.path.exists(path):
        pytest.skip("dual_epinion2_full_epoch.npz not minted")
    g = golden("dual_epinion2_full_epoch")
    raw_train, raw_test = _epinion2_trust_raw(golden)
    args, dataset, net = _dual_epinion2(data_root)
    td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2, test2 = Data(raw_train, dataset.n_users, shuffle=False), Data(raw_test, dataset.n_users, shuffle=False, test=True)
    cap = 3 * int(g["trust_batch_size"])
    net = net.to(DEV)
    st = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr, deterministic=True)
    cum, n_paths = [], []
    totals = train_epoch_dual(st, td, train2, by_user, cap, cum_every=100, cum_out=cum, n_paths_out=n_paths).cpu().numpy()
    assert st.t == int(g["n_steps"]) and np.array_equal(np.asarray(n_paths), g["n_paths"])
    cum = torch.stack(cum).cpu().numpy()
    dev_cum = (np.abs(cum[:, 0] - g["loss1_cum"]) / g["loss1_cum"]).max(), (np.abs(cum[:, 1] - g["loss2_cum"]) / g["loss2_cum"]).max()
    net.eval()
    with torch.no_grad():
        ret = rec_test(net, dataset.testRatings, dataset.testNegatives)
        tr5 = np.asarray(trust_test5(net, test2))
    tag = "ckptend"
    par = {}
    for name, p in net.named_parameters():
        want = g[f"{tag}_state_" + name.replace(".", "__")].reshape(p.shape)
        got = p.detach().cpu().numpy()
        if name.endswith(".a"):                                   # attention vectors [2H, 1]: compare the a2 halves only
            got, want = got[got.shape[0] // 2:], want[want.shape[0] // 2:]
        par[name] = rel_err(got, want)
    dev = dict(loss1=abs(totals[0] - float(g["loss1"])) / float(g["loss1"]), loss2=abs(totals[1] - float(g["loss2"])) / float(g["loss2"]),
               cum1=float(dev_cum[0]), cum2=float(dev_cum[1]),
               task_w=float(np.abs(net.task_weights.detach().cpu().numpy() - g["task_weights"]).max()),
               rec=float(max(np.abs(ret["recall"] - g["rec_recall"]).max(), np.abs(ret["ndcg"] - g["rec_ndcg"]).max())),
               trust=float(np.abs(tr5 - g["trust"]).max()), params=max(par.values()))
    print("deterministic dual-task FULL epoch (4 906 steps), deviation from the reference's run:", dev, "worst parameter:",
          max(par, key=par.get))
    # measured on the MI355X (fixed numbers: the step is deterministic): loss sums 1.2e-6 / 2.3e-6 (running sums <= 1.0e-5), task
    # weights 1.5e-5, rec HR / NDCG 2.7e-6 (no user of 3 185 changes rank), trust HR / NDCG 1.3e-3 = one or two of the 1 024 kept
    # test paths across a top-K boundary after 4 906 steps (the metric's granularity is 9.8e-4; with the round-2 summation order of
    # the logits it was 1.0e-3: which path flips depends on the last bits); at the reference's own end-of-epoch parameters both
    # evaluations agree to 1e-4 (below).  The a1 halves of the attention vectors are not compared: A.a1 cancels in the two-way
    # softmax, their true gradient is zero, the reference's is 1e-8 noise that Adam turns into a random walk.
    assert dev["loss1"] <= 5e-5 and dev["loss2"] <= 5e-5 and dev["cum1"] <= 5e-5 and dev["cum2"] <= 5e-5, dev
    assert dev["task_w"] <= 5e-5 and dev["rec"] <= 1e-4 and dev["trust"] <= 2.5 / len(raw_test[0]), dev
    # teacher forcing at the reference's own end-of-epoch parameters: both evaluations
    with torch.no_grad():
        for name, p in net.named_parameters():
            p.copy_(torch.from_numpy(g[f"{tag}_state_" + name.replace(".", "__")]).reshape(p.shape))
    net._cache = None
    net.eval()
    with torch.no_grad():
        ret = rec_test(net, dataset.testRatings, dataset.testNegatives)
        assert max(np.abs(ret["recall"] - g["rec_recall"]).max(), np.abs(ret["ndcg"] - g["rec_ndcg"]).max()) <= 1e-4
        assert np.abs(np.asarray(trust_test5(net, test2)) - g["trust"]).max() <= 1e-4


def test_fixed_task_weights_step_matches_main_11(data_root, golden):
    """main_11.py — the second dual-task driver SURVEY 2 lists: loss = loss1 + loss2 (:69), at most trust_batch_size paths per
    batch (:58-59) — through the one-call step with SPEX_STEP_FIXED_TASK_WEIGHTS (deterministic accumulation): the reference's
    first 300 steps on Epinion2 (oracle/gen_golden.py --stage epochs-dual11-epinion2): path counts, per-step losses of the first
    16 steps, running loss sums, both tasks' metrics, trained tables; task_weights stay at their initial zeros."""
    from collections import defaultdict
    import utility1.dataloader as dl
    from utility1.batch_test import rec_test
    from utility2.batch_test_gnn import trust_test5
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper, train_epoch_dual
    g = golden("dual11_epinion2_epochs")
    raw_train, raw_test = _epinion2_trust_raw(golden)
    args, dataset, net = _dual_epinion2(data_root)
    td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2, test2 = Data(raw_train, dataset.n_users, shuffle=False), Data(raw_test, dataset.n_users, shuffle=False, test=True)
    cap = int(g["trust_batch_size"])
    net = net.to(DEV)
    st = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr, deterministic=True, fixed_task_weights=True)
    n_steps = int(g["n_steps"])
    cum, n_paths = [], []
    train_epoch_dual(st, td, train2, by_user, cap, max_steps=n_steps, cum_every=100, cum_out=cum, n_paths_out=n_paths)
    assert st.t == n_steps and np.array_equal(np.asarray(n_paths), g["n_paths"])
    cum = torch.stack(cum).cpu().numpy()
    assert (np.abs(cum[:, 0] - g["loss1_cum"]) / g["loss1_cum"]).max() <= 2e-5 and (np.abs(cum[:, 1] - g["loss2_cum"]) / g["loss2_cum"]).max() <= 2e-5
    assert torch.equal(net.task_weights.detach().cpu(), torch.zeros(2)) and np.array_equal(g["task_weights"], np.zeros(2, np.float32))
    net.eval()
    with torch.no_grad():
        ret = rec_test(net, dataset.testRatings, dataset.testNegatives)
        assert max(np.abs(ret["recall"] - g["rec_recall"]).max(), np.abs(ret["ndcg"] - g["rec_ndcg"]).max()) <= 1e-4
        assert np.abs(np.asarray(trust_test5(net, test2)) - g["trust"]).max() <= 1e-4
    uw, iw = net.embedding_user.weight.detach().cpu().numpy(), net.embedding_item.weight.detach().cpu().numpy()
    tab = dict(user=rel_err(uw[g["rows_u"]], g["user_w"]), item=rel_err(iw[g["rows_i"]], g["item_w"]), w=rel_err(net.w.detach().cpu().numpy(), g["w"]))
    # the same 300 steps through the unchanged-driver loop (autograd path, fixed summation orders): main_11.py:50-72 line by line
    import random
    from torch.utils.data import DataLoader
    from spex_amd import ops
    args, dataset, ref = _dual_epinion2(data_root)
    loader = DataLoader(dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat), batch_size=256, shuffle=True)
    ref = ref.to(DEV)
    opt = torch.optim.Adam(ref.parameters(), lr=args.lr)
    ops.set_deterministic(True)
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        loader.dataset.ng_sample()
        ref.train()
        for step, (user, item, label) in enumerate(loader):
            if step == n_steps:
                break
            opt.zero_grad()
            chosen = []
            for u in set(user.numpy().tolist()):
                chosen.extend(by_user[u])
            if len(chosen) > cap:
                chosen = random.sample(chosen, cap)
            l1, l2 = ref(users=user.to(DEV), items=item.to(DEV), labels=label.to(DEV), slice_indices=np.array(chosen, dtype=int),
                         trust_data=train2, flag=0)
            (l1 + l2).backward()
            opt.step()
    finally:
        ops.set_deterministic(False)
        torch.use_deterministic_algorithms(False)
    ru, ri = ref.embedding_user.weight.detach().cpu().numpy(), ref.embedding_item.weight.detach().cpu().numpy()
    tab_ref = dict(user=rel_err(ru[g["rows_u"]], g["user_w"]), item=rel_err(ri[g["rows_i"]], g["item_w"]), w=rel_err(ref.w.detach().cpu().numpy(), g["w"]))
    both = dict(user=rel_err(uw, ru), item=rel_err(iw, ri))
    # measured: one-call step 4.7e-5 / 4.3e-4 (user / item rows), driver loop 7.9e-5 / 6.7e-4, the two paths 1.6e-4 apart — item
    # rows that only see 3-hop gradients of ~1e-9 move by Adam's lr whatever the gradient's size, so their early steps follow the
    # sign of rounding noise in every implementation; the function (losses 2e-5, HR / NDCG 1e-4 above) is what is pinned
    print("main_11, 300 steps, trained tables vs the reference's: one-call step", tab, "driver loop", tab_ref, "step vs loop", both)
    assert max(tab.values()) <= 1e-3 and max(tab_ref.values()) <= 1e-3, (tab, tab_ref)
