"""BASELINE config 4 (NGCF) on the GPU: the fused layer kernels against a torch restatement of the same layer (autograd
gives the gradients), the drop-in `utility.load_data` / `utility.batch_test` modules and the whole training run
against goldens minted from the reference (oracle/gen_golden.py --stage ngcf-epochs / ngcf-epochs-epinion2), and a
main_rec.py-shaped driver through the launcher."""
import argparse
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def ngcf_args(**kw):
    a = dict(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")
    a.update(kw)
    return argparse.Namespace(**a)


# ---------------------------------------------------------------------------------------------- the layer kernels
@pytest.mark.parametrize("n,p_drop,with_next,with_direct,pad_row", [
    (1000, 0.0, False, True, -1), (1000, 0.1, False, True, 300), (333, 0.25, True, False, -1), (16, 0.1, True, True, 3),
    (15593, 0.1, False, True, 3185)])
def test_layer_forward_and_backward_kernels_vs_torch_autograd(oracle, n, p_drop, with_next, with_direct, pad_row):
    """spex_ngcf_layer_fwd_f32 / _bwd_f32 on random inputs vs the same layer in torch fp64 with the SAME dropout mask
    (the counter-based mask restated in NumPy): outputs, both input gradients, all four weight gradients.  Upstream
    gradients are non-zero on a sparse set of rows only (as after a batch) plus one dense stretch, so both the
    skip-tile path and the arithmetic path run."""
    from spex_amd import ops
    rng = np.random.default_rng(n)
    ego = rng.normal(size=(n, 64)).astype(np.float32) * 0.3
    side = rng.normal(size=(n, 64)).astype(np.float32) * 0.3
    W_gc, W_bi = (rng.normal(size=(64, 64)).astype(np.float32) * 0.2 for _ in range(2))
    b_gc, b_bi = (rng.normal(size=64).astype(np.float32) * 0.1 for _ in range(2))
    g_all = np.zeros((n, 128), np.float32)
    rows = rng.choice(n, min(n, 96), replace=False)
    g_all[rows] = rng.normal(size=(len(rows), 128)).astype(np.float32)
    g_all[: min(n, 40)] = rng.normal(size=(min(n, 40), 128)).astype(np.float32)
    g_next = None
    if with_next:
        g_next = np.zeros((n, 64), np.float32)
        g_next[rows[:20]] = rng.normal(size=(min(20, len(rows)), 64)).astype(np.float32)
    if not with_direct:
        g_all[:, :64] = 0.0                                             # layer > 0 gets no direct term
    seed, step, layer = 0x1234567890ABCDEF, 7, 0 if with_direct else 1
    # mask in the kernels' numbering: rows above pad_row count one less
    n_mask_rows = n - (1 if 0 <= pad_row < n else 0)
    keep_ref = oracle.message_keep_mask(n_mask_rows, 64, p_drop, seed, step, layer) if p_drop > 0 else None
    keep = None
    if keep_ref is not None:
        keep = keep_ref if not (0 <= pad_row < n) else np.concatenate([keep_ref[: pad_row + 1], keep_ref[pad_row:]])
        # (the pad row itself shares a mask row with its neighbour; its values are never read)
    # ---- torch fp64 reference with autograd
    T = lambda a: torch.from_numpy(a).double().requires_grad_(True)
    te, ts, tWg, tbg, tWb, tbb = T(ego), T(side), T(W_gc), T(b_gc), T(W_bi), T(b_bi)
    out_n, e1 = oracle.ngcf_layer_torch(te, ts, tWg, tbg, tWb, tbb, None if keep is None else torch.from_numpy(keep), p_drop)
    loss = (out_n * torch.from_numpy(g_all[:, 64:]).double()).sum()
    if g_next is not None:
        loss = loss + (e1 * torch.from_numpy(g_next).double()).sum()
    loss.backward()
    # ---- kernels
    d_ego, d_side = t(ego), t(side)
    out = torch.zeros(n, 128 if with_direct else 192, device=DEV)
    e1_out = torch.empty(n, 64, device=DEV)
    drop = (p_drop, seed, step) if p_drop > 0 else None
    ops.ngcf_layer_fwd(d_ego, d_side, t(W_gc), t(b_gc), t(W_bi), t(b_bi), out, layer, with_direct, e1_out, drop=drop,
                       pad_row=pad_row)
    sl = slice(64 * (layer + 1), 64 * (layer + 2))
    assert rel_err(out[:, sl].cpu().numpy(), out_n.detach().numpy()) <= 2e-6
    assert rel_err(e1_out.cpu().numpy(), e1.detach().numpy()) <= 2e-6
    if with_direct:
        assert np.array_equal(out[:, :64].cpu().numpy(), ego)
    g_full = torch.zeros(n, out.shape[1], device=DEV)
    g_full[:, sl] = t(g_all[:, 64:])
    if with_direct:
        g_full[:, :64] = t(g_all[:, :64])
    g_side, g_ego = torch.full((n, 64), 7.0, device=DEV), torch.full((n, 64), 7.0, device=DEV)
    gW_gc, gW_bi = torch.zeros(64, 64, device=DEV), torch.zeros(64, 64, device=DEV)
    gb_gc, gb_bi = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    ops.ngcf_layer_bwd(d_ego, d_side, t(W_gc), t(b_gc), t(W_bi), t(b_bi), g_full, layer, None if g_next is None else t(g_next),
                       g_side, g_ego, gW_gc, gb_gc, gW_bi, gb_bi, drop=drop, pad_row=pad_row)
    want_gego = te.grad.numpy() + (g_all[:, :64] if with_direct else 0.0)
    assert rel_err(g_side.cpu().numpy(), ts.grad.numpy()) <= 1e-5
    assert rel_err(g_ego.cpu().numpy(), want_gego) <= 1e-5
    for got, want, nm in ((gW_gc, tWg.grad, "W_gc"), (gW_bi, tWb.grad, "W_bi"), (gb_gc, tbg.grad, "b_gc"), (gb_bi, tbb.grad, "b_bi")):
        assert rel_err(got.cpu().numpy(), want.numpy()) <= 2e-5, nm


# ---------------------------------------------------------------------------------------------- model vs G7 (two layers too)
def test_two_layer_model_gradients_vs_torch_ops_on_the_same_weights(golden):
    """layer_size [64, 64]: the fused path (NGCFPropagate: g_next chaining, slices of the concatenated table) against the
    torch-op path of the same module on the same graph (eval mode: no dropout)."""
    import scipy.sparse as sp
    from spex_amd.ngcf import NGCF
    g = golden("ngcf_tiny")
    nu, ni = int(g["n_users"]), int(g["n_items"])
    adj = sp.csr_matrix((g["val"], g["col"], g["rowptr"]), shape=(nu + ni, nu + ni))
    torch.manual_seed(5)
    m = NGCF({"n_users": nu, "n_items": ni, "norm_adj": adj}, DEV, ngcf_args(layer_size="[64,64]", mess_dropout="[0.0,0.0]")).to(DEV)
    bu, bi, bl = (torch.from_numpy(g[k]) for k in ("batch_users", "batch_items", "batch_labels"))
    m.train()
    loss = m(bu, bi, bl, flag=0)
    loss.backward()
    got = {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
    m.zero_grad()
    m._fused_ok = lambda: False                                         # same module, torch-op layers
    loss2 = m(bu, bi, bl, flag=0)
    loss2.backward()
    assert abs(loss.item() - loss2.item()) <= 2e-6
    # Both sides are fp32 sums over all rows in different association orders (kernel: per-workgroup partial sums + float
    # atomics, run-to-run order; torch: rocBLAS / reduce kernels) — measured 0.9e-5 .. 2.1e-5 between runs on the bias
    # gradients, whose row sums cancel.  The tight gate (each gradient vs fp64 autograd, <= 2e-5) is the layer test above.
    for k, p in m.named_parameters():
        assert rel_err(got[k], p.grad.cpu().numpy()) <= 5e-5, k


# ---------------------------------------------------------------------------------------------- data + eval + whole run
def _materialise(root, name, train_pairs, test_pos, test_neg):
    rec = os.path.join(root, name, "rec")
    os.makedirs(rec, exist_ok=True)
    order = np.argsort(train_pairs[:, 0], kind="stable")
    pairs = train_pairs[order]
    with open(os.path.join(rec, "train.txt"), "w") as f:
        users, start = np.unique(pairs[:, 0], return_index=True)
        for k, u in enumerate(users):
            end = start[k + 1] if k + 1 < len(users) else len(pairs)
            f.write(str(u) + "".join(" %d" % i for i in pairs[start[k]:end, 1]) + "\n")
    with open(os.path.join(rec, "test.txt"), "w") as f:
        for u, p in test_pos:
            f.write("%d %d\n" % (u, p))
    with open(os.path.join(rec, "negative.txt"), "w") as f:
        for u, negs in test_neg:
            f.write(str(u) + "".join(" %d" % i for i in negs) + "\n")
    return os.path.join(root, "")


@pytest.fixture(scope="module")
def ngcf_data_root(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("ngcf_data"))
    g = np.load(os.path.join(REPO, "tests", "golden", "ngcf_small_epochs.npz"))
    _materialise(root, "small", g["train_pairs"], list(enumerate(g["test_pos"])), list(enumerate(g["test_neg"])))
    e = np.load(os.path.join(REPO, "tests", "golden", "epinion2_dataset.npz"))
    # NGCF's files for Epinion2 are the same interactions in its own format (data_process_rec.py:277-318)
    _materialise(root, "epinion2", e["train"].astype(np.int64), list(zip(e["test_users"].astype(int), e["test_pos"].astype(int))),
                 list(zip(e["test_users"].astype(int), e["test_neg"].astype(np.int64))))
    return os.path.join(root, "")


def test_ngcf_driver_runs_through_the_launcher(ngcf_data_root):
    """tests/drivers/ngcf_driver.py — NGCF_SPEX/code/main_rec.py's imports and call sequence with the one line a maintainer
    changes (Model_Wrapper from spex_amd.ngcf) — run as INTEGRATION.md says, `python -m spex_amd.dropin <driver> ...`, three
    epochs on the 300-user graph: the printed loss sums and HR / NDCG are the reference's own run's (G12-NGCF, same seed;
    message dropout regenerated from the counter-based mask)."""
    import re
    g = np.load(os.path.join(REPO, "tests", "golden", "ngcf_small_epochs.npz"))
    script = os.path.join(REPO, "tests", "drivers", "ngcf_driver.py")
    out = subprocess.run([sys.executable, "-m", "spex_amd.dropin", script, "--data_path", ngcf_data_root, "--dataset", "small",
                          "--epoch", "3"], cwd=REPO, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) == 3, out.stdout[-2000:]
    for ep, line in enumerate(lines):
        nums = [float(x) for x in re.findall(r"-?\d+\.\d+", line)]
        loss, rec, ndcg = nums[0], np.array(nums[1:4]), np.array(nums[4:7])
        assert abs(loss - g["losses"][ep]) <= 1e-4 * g["losses"][ep] + 2e-5, (ep, loss, g["losses"][ep])
        assert np.abs(rec - g["recall"][ep]).max() <= 2e-4 and np.abs(ndcg - g["ndcg"][ep]).max() <= 2e-4, (ep, rec, g["recall"][ep])


def test_unchanged_ngcf_driver_runs_on_the_hip_spmm_through_the_operator_hook(ngcf_data_root):
    """tests/drivers/ngcf_unchanged_driver.py keeps the model class IN the driver, like NGCF_SPEX/code/main_rec.py:36-113, and calls
    `torch.sparse.mm(self.norm_adj.to(self.device), ego)` (:76) — it names nothing of libspexhip.  Run through the launcher, the
    operator hook (spex_amd/dropin/sparse_hook.py) recognises the adjacency the drop-in Data handed out, drops the per-call
    upload and runs the product and its autograd backward on spex_spmm_f32: EVERY product of the run goes through the hook (none
    falls back to ATen), and the three epochs on the 300-user graph reproduce the reference's own run (G12-NGCF: loss sums, HR /
    NDCG per epoch; dropout masks from the goldens' counter-based generator)."""
    import ast
    import re
    g = np.load(os.path.join(REPO, "tests", "golden", "ngcf_small_epochs.npz"))
    script = os.path.join(REPO, "tests", "drivers", "ngcf_unchanged_driver.py")
    env = dict(os.environ, SPEX_TEST_COUNTER_DROPOUT=str(int(g["drop_seed"])))
    out = subprocess.run([sys.executable, "-m", "spex_amd.dropin", script, "--data_path", ngcf_data_root, "--dataset", "small",
                          "--epoch", "3"], cwd=REPO, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) == 3, out.stdout[-2000:]
    for ep, line in enumerate(lines):
        nums = [float(x) for x in re.findall(r"-?\d+\.\d+", line)]
        loss, rec, ndcg = nums[0], np.array(nums[1:4]), np.array(nums[4:7])
        assert abs(loss - g["losses"][ep]) <= 1e-4 * g["losses"][ep] + 2e-5, (ep, loss, g["losses"][ep])
        assert np.abs(rec - g["recall"][ep]).max() <= 2e-4 and np.abs(ndcg - g["ndcg"][ep]).max() <= 2e-4, (ep, rec, g["recall"][ep])
    hook = [l for l in out.stdout.splitlines() if l.startswith("sparse_hook ")]
    assert len(hook) == 1 and hook[0].split()[1] == "True", out.stdout[-800:]
    stats = ast.literal_eval(hook[0].split(" ", 2)[2])
    n_steps = int(g["n_steps"])
    assert stats["fallback_calls"] == 0 and stats["hip_calls"] == n_steps + 3, stats          # one product per step + one per test()
    assert stats["uploads_avoided"] == stats["hip_calls"] - 1, stats                          # the matrix went to the device once


def _run_reference_loop(ds, n_epochs, g, root, metric_tol=1e-4):
    """main_rec.py:116-148 on the drop-in modules: Data, NGCF, torch Adam, the DataLoader of load_train_data, test()."""
    from spex_amd.dropin.ngcf.utility import batch_test
    from spex_amd.dropin.ngcf.utility.load_data import Data
    from spex_amd.ngcf import NGCF
    torch.manual_seed(int(g["seed"]))
    random.seed(int(g["seed"]))
    np.random.seed(int(g["seed"]))
    data = Data(path=root + ds, batch_size=256)
    batch_test.use_data(data)
    assert (data.n_users, data.n_items, data.n_train) == (int(g["n_users"]), int(g["n_items"]), int(g["n_train"]))
    plain, norm, mean = data.get_adj_mat()
    p = [float(x) for x in g["mess_dropout"]]
    model = NGCF({"n_users": data.n_users, "n_items": data.n_items, "norm_adj": norm}, DEV,
                 ngcf_args(mess_dropout=str(p))).to(DEV)
    model.message_dropout_seed = int(g["drop_seed"])
    for k in g.files:                                                     # same initial parameters for the same seed
        if k.startswith("init_") and not k.endswith("_sha"):
            name = k[5:].replace("__", ".")
            assert rel_err(model.state_dict()[name].cpu().numpy(), g[k]) == 0.0, name
    opt = torch.optim.Adam(model.parameters(), lr=float(g["lr"]))
    step = 0
    eval_at = {int(k): v for k, v in zip(g["eval_steps"], g["eval_metrics"])} if "eval_steps" in g.files else {}
    drift = {}

    def maybe_eval():
        if step in eval_at:
            r = batch_test.test(model, list(data.test_set.keys()), drop_flag=True)
            drift[step] = float(np.abs(np.concatenate([r["recall"], r["ndcg"]]) - eval_at[step]).max())
    maybe_eval()
    for epoch in range(n_epochs):
        loader = data.load_train_data()
        total = 0.0
        for k, (user, item, labels) in enumerate(loader):
            if step == 0:
                assert np.array_equal(torch.stack([user, item, labels.long()]).numpy(), g["first_batch"])
            model.train()
            opt.zero_grad()
            loss = model(user=user.to(DEV), item=item.to(DEV), labels_list=labels.to(DEV), flag=0)
            loss.backward()
            opt.step()
            li = loss.item()
            if step < len(g["step_losses"]):
                assert abs(li - g["step_losses"][step]) <= 2e-5 * max(1.0, abs(g["step_losses"][step])), (step, li)
            total += li
            step += 1
            maybe_eval()
        assert abs(total - g["losses"][epoch]) <= 1e-4 * g["losses"][epoch], (epoch, total, g["losses"][epoch])
        ret = batch_test.test(model, list(data.test_set.keys()), drop_flag=True)
        drift[("epoch", epoch)] = float(max(np.abs(ret["recall"] - g["recall"][epoch]).max(), np.abs(ret["ndcg"] - g["ndcg"][epoch]).max()))
    print("metric drift vs the reference run:", drift)
    for key, dv in drift.items():
        assert dv <= metric_tol, (key, dv)
    assert step == int(g["n_steps"])
    return model


def test_whole_ngcf_run_matches_the_reference_small(golden, ngcf_data_root):
    """G12-NGCF: three epochs on the 300-user graph — same sampler stream, same first batch, same per-step losses, same
    loss sums, same recall / ndcg after every epoch, same trained weights as the reference run (message dropout on,
    masks from the shared counter-based generator)."""
    from spex_amd import ops
    g = golden("ngcf_small_epochs")
    # in the deterministic accumulation mode: the module's scoring backward then adds its per-sample rows in slot order, the dense layer
    # backward has had no float atomic since round 4, the products never had any — the run is a function of the seeds, so this
    # test either holds or does not (with the scoring's float atomics the final weights of this scale-invariant model landed on
    # either side of the 1e-4 line from run to run: ~1 in 4 runs failed at 9e-4)
    ops.set_deterministic(True)
    try:
        model = _run_reference_loop("small", 3, g, ngcf_data_root)
    finally:
        ops.set_deterministic(False)
    sd = model.state_dict()
    assert rel_err(sd["user_embedding.weight"].cpu().numpy(), g["user_w"]) <= 5e-5
    assert rel_err(sd["item_embedding.weight"].cpu().numpy(), g["item_w"]) <= 5e-5
    for k in g.files:
        if k.startswith("final_"):
            assert rel_err(sd[k[6:].replace("__", ".")].cpu().numpy(), g[k]) <= 1e-4, k


def test_ngcf_module_run_repeats_bit_for_bit_in_the_deterministic_mode(golden, ngcf_data_root):
    """The three epochs of the small run through the MODULE (autograd over the HIP kernels, torch Adam) twice under ops.set_deterministic:
    identical parameters, bit for bit — the scoring backward adds its per-sample rows in slot order (every 64-column block of NGCF's
    concatenated table), the dense layer backward adds its workgroups' weight-gradient blocks in block order (no float atomic since
    round 4), the SpMM products are one fmaf chain per element."""
    from spex_amd import ops
    g = golden("ngcf_small_epochs")
    ops.set_deterministic(True)
    try:
        a = {k: v.clone() for k, v in _run_reference_loop("small", 3, g, ngcf_data_root).state_dict().items()}
        b = _run_reference_loop("small", 3, g, ngcf_data_root).state_dict()
    finally:
        ops.set_deterministic(False)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_native_ngcf_epoch_and_overlapped_sampling_equal_the_python_loop(golden, ngcf_data_root):
    """trainer.train_epoch_ngcf issues the single-layer model's epoch as ONE native call (spex_ngcf_epoch_bce_f32: train() of
    NGCF_SPEX/code/main_rec.py:116-131, batch after batch) where nothing has to happen on the host between two steps; asking for
    per-step losses keeps the Python loop; trainer.train_epochs_ngcf prepares the next epoch's samples (Data.sample_epoch: the blocked
    replay of the `random` stream) and shuffle on a second thread meanwhile.  Three epochs on the 300-user graph in the DETERMINISTIC
    step, where a run is a pure function of its inputs: the three ways end in bit-identical parameters and loss sums."""
    from spex_amd.dropin.ngcf.utility.load_data import Data
    from spex_amd.ngcf import NGCF
    from spex_amd.trainer import NGCFStepper, train_epoch_ngcf, train_epochs_ngcf
    g = golden("ngcf_small_epochs")
    out = []
    for way in ("native", "python", "overlapped"):
        torch.manual_seed(int(g["seed"])); random.seed(int(g["seed"])); np.random.seed(int(g["seed"]))
        data = Data(path=ngcf_data_root + "small", batch_size=256)
        _, norm, _ = data.get_adj_mat()
        model = NGCF({"n_users": data.n_users, "n_items": data.n_items, "norm_adj": norm}, DEV,
                     ngcf_args(mess_dropout=str([float(x) for x in g["mess_dropout"]]))).to(DEV)
        model.message_dropout_seed = int(g["drop_seed"])
        model.train()
        st = NGCFStepper(model, lr=float(g["lr"]), deterministic=True)
        if way == "overlapped":
            totals = train_epochs_ngcf(st, data, 3)
        else:
            totals = [train_epoch_ngcf(st, data, step_losses=[] if way == "python" else None).item() for _ in range(3)]
        assert st.t == int(g["n_steps"]) and model.dropout_step == int(g["n_steps"])
        out.append((totals, [p.detach().clone() for p in model.parameters()]))
        for ep in range(3):
            assert abs(totals[ep] - g["losses"][ep]) <= 5e-5 * g["losses"][ep]
    for totals, params in out[1:]:
        assert np.allclose(totals, out[0][0], rtol=1e-6, atol=0)
        assert all(torch.equal(a, b) for a, b in zip(params, out[0][1]))


def test_on_device_ngcf_epochs_match_the_reference_small(golden, ngcf_data_root):
    """The same golden through the autograd-free loop: spex_amd.trainer.NGCFStepper (a fixed sequence of launches per
    step: SpMM, fused layer, scoring, fused layer backward, SpMM^T, two Adam passes) driven by train_epoch_ngcf (the
    reference's sampler stream and the DataLoader's shuffle order, replayed) — same loss sums and metrics per epoch."""
    from spex_amd.dropin.ngcf.utility import batch_test
    from spex_amd.dropin.ngcf.utility.load_data import Data
    from spex_amd.ngcf import NGCF
    from spex_amd.trainer import NGCFStepper, train_epoch_ngcf
    g = golden("ngcf_small_epochs")
    torch.manual_seed(int(g["seed"])); random.seed(int(g["seed"])); np.random.seed(int(g["seed"]))
    data = Data(path=ngcf_data_root + "small", batch_size=256)
    batch_test.use_data(data)
    _, norm, _ = data.get_adj_mat()
    model = NGCF({"n_users": data.n_users, "n_items": data.n_items, "norm_adj": norm}, DEV,
                 ngcf_args(mess_dropout=str([float(x) for x in g["mess_dropout"]]))).to(DEV)
    model.message_dropout_seed = int(g["drop_seed"])
    st = NGCFStepper(model, lr=float(g["lr"]))
    for epoch in range(3):
        model.train()
        total = train_epoch_ngcf(st, data).item()
        assert abs(total - g["losses"][epoch]) <= 5e-5 * g["losses"][epoch], (epoch, total, g["losses"][epoch])
        ret = batch_test.test(model, list(data.test_set.keys()), drop_flag=True)      # the module sees the trained weights
        assert np.abs(ret["recall"] - g["recall"][epoch]).max() <= 1e-4 and np.abs(ret["ndcg"] - g["ndcg"][epoch]).max() <= 1e-4
    assert st.t == int(g["n_steps"]) and model.dropout_step == int(g["n_steps"])


def _ngcf_epinion2_model(g, root, layer_size="[64]"):
    from spex_amd.dropin.ngcf.utility import batch_test
    from spex_amd.dropin.ngcf.utility.load_data import Data
    from spex_amd.ngcf import NGCF
    torch.manual_seed(int(g["seed"])); random.seed(int(g["seed"])); np.random.seed(int(g["seed"]))
    data = Data(path=root + "epinion2", batch_size=256)
    batch_test.use_data(data)
    _, norm, _ = data.get_adj_mat()
    model = NGCF({"n_users": data.n_users, "n_items": data.n_items, "norm_adj": norm}, DEV,
                 ngcf_args(mess_dropout=str([float(x) for x in g["mess_dropout"]]), layer_size=layer_size)).to(DEV)
    model.message_dropout_seed = int(g["drop_seed"])
    return data, model, batch_test


def test_ngcf_first_300_steps_with_the_references_own_dropout_noise(golden, ngcf_data_root):
    """G12-NGCF with NOTHING substituted in the model: 300 steps of NGCF_SPEX/code/main_rec.py on Epinion2 minted with the
    reference's nn.Dropout modules left alone (oracle/gen_golden.py --stage ngcf-native-dropout-epinion2: their noise is
    at::dropout's empty_like(x).bernoulli_(1 - p) from torch's global generator, one [N, 64] draw per step).  The one-call step
    replays that draw on the host at the same point of the stream and hands the bytes to the kernels (spex_ngcf_message_mask), so
    the run drops the same activations: every step's loss, then test().  (The other NGCF goldens inject a counter-based mask INTO
    the reference's model; this one pins the path against the unmodified model.)"""
    from spex_amd.trainer import NGCFStepper, train_epoch_ngcf
    g = golden("ngcf_epinion2_native_dropout")
    data, model, batch_test = _ngcf_epinion2_model(g, ngcf_data_root)
    st = NGCFStepper(model, lr=float(g["lr"]), deterministic=True)
    st.dropout_stream = "reference"
    n_steps = int(g["n_steps"])
    step_losses = []
    model.train()
    total = train_epoch_ngcf(st, data, step_losses=step_losses, max_steps=n_steps).item()
    assert len(step_losses) == n_steps == len(g["step_losses"])
    dev = np.abs(np.asarray(step_losses) - g["step_losses"])
    print("NGCF, reference's own dropout noise, %d steps: max per-step loss deviation %.2e, loss sum %.6f vs %.6f"
          % (n_steps, dev.max(), total, float(g["losses"][0])))
    # (300 steps is the window in which two fp32 runs of this model stay together: a 1 500-step mint replayed the same way drifts
    #  8e-5 / 2e-3 / 5e-3 per 300-step window while its loss SUM still agrees to 1.8e-5 — the chaotic growth the reference's own two
    #  mints show, not a desynchronised mask stream, which would jump at once)
    assert dev.max() <= 5e-6, (int(dev.argmax()), float(dev.max()))           # measured on the MI355X: 6.0e-7 over the 300 steps
    assert abs(total - g["losses"][0]) <= 2e-6 * g["losses"][0]              # (2e-8)
    ret = batch_test.test(model, list(data.test_set.keys()), drop_flag=True)
    got = np.concatenate([ret["recall"], ret["ndcg"]])
    want = np.concatenate([g["recall"][0], g["ndcg"][0]])
    print("HR / NDCG deviation after %d steps: %.2e" % (n_steps, np.abs(got - want).max()))
    assert np.abs(got - want).max() <= 1e-4, (got, want)


def test_ngcf_module_under_the_references_loop_with_its_own_dropout_noise(golden, ngcf_data_root):
    """The same golden through the MODULE (spex_amd.ngcf.NGCF: autograd over the HIP kernels) under main_rec.py:116-148's own
    loop — DataLoader(shuffle=True), torch Adam, zero_grad / backward / step — with `model.dropout_stream = "reference"`: the
    module draws nn.Dropout's noise where the reference does, so nothing but the import line differs from the reference's run.
    First 64 steps' losses (the shuffle and the per-step noise both come off torch's global generator: a desynchronised stream
    would show at step 0 or 1)."""
    g = golden("ngcf_epinion2_native_dropout")
    data, model, _ = _ngcf_epinion2_model(g, ngcf_data_root)
    model.dropout_stream = "reference"
    opt = torch.optim.Adam(model.parameters(), lr=float(g["lr"]))
    model.train()
    losses = []
    for k, (user, item, labels) in enumerate(data.load_train_data()):
        if k == 64:
            break
        opt.zero_grad()
        loss = model(user=user.to(DEV), item=item.to(DEV), labels_list=labels.to(DEV), flag=0)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    dev = np.abs(np.asarray(losses) - g["step_losses"][:64])
    print("NGCF module, reference loop + reference noise, 64 steps: max per-step loss deviation %.2e" % dev.max())
    assert dev.max() <= 5e-6, (int(dev.argmax()), float(dev.max()))


def _unit(w):
    w = np.asarray(w, np.float64)
    return w / np.sqrt((w ** 2).sum())


def test_whole_ngcf_epoch_matches_the_reference_epinion2(golden, ngcf_data_root):
    """One full NGCF epoch on Epinion2 (BASELINE config 4's size: 4 757 Adam steps, message dropout on) + test() at steps 0 /
    500 / 1 500 / end, through the on-device loop with the DETERMINISTIC step (trainer.train_epoch_ngcf +
    NGCFStepper(deterministic=True)): the run is a pure function of the seeds — it repeats bit for bit
    (tests/test_gpu_deterministic.py) — so its deviation from the reference's run is a FIXED number, not a distribution.

    What can be expected of that number: the REFERENCE'S OWN run is not reproducible (torch's CPU index_put_(accumulate) and
    threaded reductions; Adam amplifies the last bits along NGCF's scale-invariant weight direction) — two mints of the same
    reference run (ngcf_epinion2_ref_spread.npz) differ by 1.0e-4 in the epoch's loss sum, by 4.5e-4 / 1.9e-3 / 8.2e-4 in
    HR / NDCG at steps 500 / 1 500 / end, by 0.39 / 0.22 in the DIRECTION of W_gc / W_bi and by 0.22 / 0.12 in the table rows.
    A trajectory can therefore be pinned no tighter than that (measured for this build's deterministic run against mint b:
    1.1e-4, 3.1e-4 / 2.2e-3 / 1.9e-3, 0.50 / 0.24, 0.17 / 0.18 — as far from either mint as they are from each other); what IS
    pinned to the north-star gates is the FUNCTION at the reference's own trained weights (test_ngcf_teacher_forced_checkpoints:
    loss 2e-5, gradients 5e-5, HR / NDCG 1e-4 at steps 500 / 1 500 / end), the first 32 per-step losses (2e-5) and the
    evaluation at step 0 (1e-4).  Every trajectory gate below is TWICE THE REFERENCE'S OWN run-to-run spread in the same
    measure (for the ranking metrics: of the largest spread it shows at any evaluated step — the drift is not monotone)."""
    from spex_amd.trainer import NGCFStepper, train_epoch_ngcf
    g, sp = golden("ngcf_epinion2_epochs"), golden("ngcf_epinion2_ref_spread")
    ref_loss_spread = abs(sp["losses_a"][0] - sp["losses_b"][0]) / sp["losses_b"][0]
    ref_metric_spread = {int(k): float(v) for k, v in zip(sp["eval_steps"], np.abs(sp["eval_metrics_a"] - sp["eval_metrics_b"]).max(1))}
    ref_end_spread = float(max(np.abs(sp["recall_a"] - sp["recall_b"]).max(), np.abs(sp["ndcg_a"] - sp["ndcg_b"]).max()))
    ref_metric_max = max(max(ref_metric_spread.values()), ref_end_spread)
    ref_dir = {k: float(np.abs(_unit(sp["final_" + k + "_a"]) - _unit(sp["final_" + k + "_b"])).max() / np.abs(_unit(sp["final_" + k + "_b"])).max())
               for k in ("GC_Linear_list__0__weight", "Bi_Linear_list__0__weight")}
    ref_rows = {"user_rows": rel_err(sp["user_w_a"], sp["user_w_b"]), "item_rows": rel_err(sp["item_w_a"], sp["item_w_b"])}
    data, model, batch_test = _ngcf_epinion2_model(g, ngcf_data_root)
    users_to_test = list(data.test_set.keys())
    for k in g.files:                                                     # same initial parameters for the same seed
        if k.startswith("init_") and not k.endswith("_sha"):
            assert rel_err(model.state_dict()[k[5:].replace("__", ".")].cpu().numpy(), g[k]) == 0.0, k
    st = NGCFStepper(model, lr=float(g["lr"]), deterministic=True)
    eval_at = {int(k): v for k, v in zip(g["eval_steps"], g["eval_metrics"])}
    drift = {}

    def make_eval(step):
        def fn():
            r = batch_test.test(model, users_to_test, drop_flag=True)
            drift[step] = float(np.abs(np.concatenate([r["recall"], r["ndcg"]]) - eval_at[step]).max())
            model.train()
        return fn
    step_losses = []
    model.train()
    total = train_epoch_ngcf(st, data, callbacks={k: make_eval(k) for k in eval_at}, step_losses=step_losses).item()
    assert st.t == int(g["n_steps"]) and model.dropout_step == int(g["n_steps"])
    first = np.abs(np.asarray(step_losses[:len(g["step_losses"])]) - g["step_losses"])
    assert first.max() <= 2e-5, (int(first.argmax()), float(first.max()))               # the function, step by step
    loss_dev = abs(total - g["losses"][0]) / g["losses"][0]
    ret = batch_test.test(model, users_to_test, drop_flag=True)
    end_dev = float(max(np.abs(ret["recall"] - g["recall"][0]).max(), np.abs(ret["ndcg"] - g["ndcg"][0]).max()))
    sd = model.state_dict()
    uw, iw = sd["user_embedding.weight"].cpu().numpy(), sd["item_embedding.weight"].cpu().numpy()
    # parameters modulo NGCF's scale direction: the layer output is invariant to a joint positive rescaling of (W_gc, b_gc, W_bi,
    # b_bi) — LeakyReLU is positively homogeneous and the output is L2-normalised — so only the DIRECTION of the weights is
    # determined by the loss; table rows are compared after normalising each row (what scoring reads through the layer)
    dirs = {k: float(np.abs(_unit(sd[k.replace("__", ".")].cpu().numpy()) - _unit(g["final_" + k])).max() / np.abs(_unit(g["final_" + k])).max())
            for k in ("GC_Linear_list__0__weight", "Bi_Linear_list__0__weight")}
    rows = {"user_rows": rel_err(uw[g["rows_u"]], g["user_w"]), "item_rows": rel_err(iw[g["rows_i"]], g["item_w"])}
    print("deterministic NGCF epoch vs the reference run: loss sum dev %.2e (reference's own spread %.2e), metric dev %s / end %.2e "
          "(reference's own %s / %.2e), weight directions %s (reference's own %s), table rows %s (reference's own %s)"
          % (loss_dev, ref_loss_spread, drift, end_dev, ref_metric_spread, ref_end_spread, dirs, ref_dir, rows, ref_rows))
    assert drift[0] <= 1e-4                                                              # the evaluation path, seeded weights
    assert loss_dev <= 2 * ref_loss_spread, (loss_dev, ref_loss_spread)
    for k, dv in drift.items():
        if k:
            assert dv <= 2 * ref_metric_max, (k, dv, ref_metric_max)
    assert end_dev <= 2 * ref_metric_max, (end_dev, ref_metric_max)
    for k in dirs:
        assert dirs[k] <= 2 * ref_dir[k], (k, dirs[k], ref_dir[k])
    for k in rows:
        assert rows[k] <= 2 * ref_rows[k], (k, rows[k], ref_rows[k])


@pytest.mark.parametrize("tag", ["ckpt500", "ckpt1500", "ckptend"])
def test_ngcf_teacher_forced_checkpoints(golden, ngcf_data_root, tag):
    """Teacher forcing at TRAINED weights: the reference's full parameter state in front of steps 500 / 1 500 and at the end of
    its Epinion2 epoch (oracle/gen_golden.py --stage ngcf-epochs-epinion2 -> ngcf_epinion2_ckpt.npz: every parameter in fp32,
    that step's batch and dropout step, its loss, its gradients — weights in full, tables as sampled rows + column sums +
    Frobenius norm —, and the reference's test() at that state).  Loaded into the GPU model: ONE forward / backward on the same
    batch with the same dropout mask -> loss <= 2e-5, gradients <= 5e-5; the same step through the one-call stepper (both
    accumulation modes) -> loss <= 2e-5; test() -> HR / NDCG @ {10, 20, 50} <= 1e-4.  Unlike a trajectory, this pins the
    function the two implementations compute after hundreds / thousands of Adam steps."""
    from spex_amd.trainer import NGCFStepper
    g = golden("ngcf_epinion2_ckpt")
    data, model, batch_test = _ngcf_epinion2_model(g, ngcf_data_root)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"{tag}_state_" + name.replace(".", "__")]))
    user, item, labels = (torch.from_numpy(g[f"{tag}_batch"][k]) for k in range(3))
    want_loss = float(g[f"{tag}_loss"])
    # ---- forward / backward through the drop-in model (autograd Functions on the HIP kernels)
    model.train()
    model.dropout_step = int(g[f"{tag}_drop_step"])
    model.zero_grad()
    loss = model(user=user.to(DEV), item=item.to(DEV), labels_list=labels.float().to(DEV), flag=0)
    loss.backward()
    assert abs(loss.item() - want_loss) <= 2e-5, (loss.item(), want_loss)
    for name, p in model.named_parameters():
        key = f"{tag}_grad_" + name.replace(".", "__")
        got = p.grad.cpu().numpy()
        if key + "_rows" in g.files:                                     # the two tables
            fro = float(g[key + "_fro"])
            assert abs(np.sqrt((got.astype(np.float64) ** 2).sum()) - fro) <= 5e-5 * fro, name
            cs = g[key + "_colsum"]
            assert np.abs(got.astype(np.float64).sum(0) - cs).max() <= 5e-5 * max(np.abs(cs).max(), 1e-6), name
            got = got[g[key + "_rows"]]
        assert rel_err(got, g[key]) <= 5e-5, (name, rel_err(got, g[key]))
    # ---- the same step through the one-call stepper, both accumulation modes (the loss is read before anything is updated)
    for det in (False, True):
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(g[f"{tag}_state_" + name.replace(".", "__")]))
        model.dropout_step = int(g[f"{tag}_drop_step"])
        st = NGCFStepper(model, lr=float(g["lr"]), deterministic=det)
        acc = torch.zeros(1, device=DEV)
        st.step(user.to(DEV), item.to(DEV), labels.float().to(DEV), loss_acc=acc)
        assert abs(acc.item() / len(user) - want_loss) <= 2e-5, (det, acc.item() / len(user), want_loss)
    # ---- evaluation at the reference's trained weights: the north-star gate
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"{tag}_state_" + name.replace(".", "__")]))
    model.eval()
    step = {"ckpt500": 500, "ckpt1500": 1500, "ckptend": int(g["n_steps"])}[tag]
    want = g["eval_metrics"][list(g["eval_steps"]).index(step)]
    ret = batch_test.test(model, list(data.test_set.keys()), drop_flag=True)
    got = np.concatenate([ret["recall"], ret["ndcg"]])
    assert np.abs(got - want).max() <= 1e-4, (got, want)


@pytest.mark.parametrize("tag", ["ckpt40", "ckptend"])
def test_two_layer_ngcf_teacher_forced_against_the_reference(golden, ngcf_data_root, tag):
    """NGCF beyond one layer (`--layer_size [64,64] --mess_dropout [0.1,0.1]`, NGCF_SPEX/code/ngcf_parser.py:12; the layer loop of
    main_rec.py:71-93) pinned against the REFERENCE: its full parameter state in front of step 40 and after 60 steps of its Epinion2
    run (oracle/gen_golden.py --stage ngcf-2layer-epinion2 -> ngcf_epinion2_2layer_ckpt.npz: that step's batch, dropout step, loss,
    every gradient — the four weight matrices of BOTH layers in full, the tables as sampled rows + column sums + Frobenius norm).
    Loaded into the GPU model: one forward / backward through the module (dense layer forward, rows-form backward of the last layer,
    push-form A^T product, the 4-wave DENSE backward of the first layer, pull-form A^T product) -> loss <= 2e-5, gradients <= 5e-5;
    the same step through NGCFStepper -> the same loss, and the parameters one Adam step later agree with the module + torch Adam;
    test() at the reference's trained 2-layer weights -> HR / NDCG <= 1e-4."""
    from spex_amd.trainer import NGCFStepper
    g = golden("ngcf_epinion2_2layer_ckpt")
    data, model, batch_test = _ngcf_epinion2_model(g, ngcf_data_root, layer_size="[64,64]")
    assert model.n_layers == 2

    def load_state():
        with torch.no_grad():
            for name, p in model.named_parameters():
                p.copy_(torch.from_numpy(g[f"{tag}_state_" + name.replace(".", "__")]))
        model.dropout_step = int(g[f"{tag}_drop_step"])
    load_state()
    user, item, labels = (torch.from_numpy(g[f"{tag}_batch"][k]) for k in range(3))
    want_loss = float(g[f"{tag}_loss"])
    model.train()
    model.zero_grad()
    loss = model(user=user.to(DEV), item=item.to(DEV), labels_list=labels.float().to(DEV), flag=0)
    loss.backward()
    assert abs(loss.item() - want_loss) <= 2e-5, (loss.item(), want_loss)
    for name, p in model.named_parameters():
        key = f"{tag}_grad_" + name.replace(".", "__")
        got = p.grad.cpu().numpy()
        if key + "_rows" in g.files:                                     # the two tables
            fro = float(g[key + "_fro"])
            assert abs(np.sqrt((got.astype(np.float64) ** 2).sum()) - fro) <= 5e-5 * fro, name
            cs = g[key + "_colsum"]
            assert np.abs(got.astype(np.float64).sum(0) - cs).max() <= 5e-5 * max(np.abs(cs).max(), 1e-6), name
            got = got[g[key + "_rows"]]
        assert rel_err(got, g[key]) <= 5e-5, (name, rel_err(got, g[key]))
    # ---- one optimiser step two ways from the same state: module + torch Adam, and the stepper
    opt = torch.optim.Adam(model.parameters(), lr=float(g["lr"]))
    opt.step()
    want = {n: p.detach().clone() for n, p in model.named_parameters()}
    load_state()
    st = NGCFStepper(model, lr=float(g["lr"]))
    acc = torch.zeros(1, device=DEV)
    st.step(user.to(DEV), item.to(DEV), labels.float().to(DEV), loss_acc=acc)
    assert abs(acc.item() / len(user) - want_loss) <= 2e-5, (acc.item() / len(user), want_loss)
    for n, p in model.named_parameters():
        # (Adam's FIRST step moves a weight by lr * g / (|g| + 1e-8): ~lr whatever the gradient's size, and for the few elements
        #  whose gradient is itself at rounding level the quotient amplifies that rounding — so the bulk is compared through the mean,
        #  the worst element only against the step size)
        dv = (p.detach() - want[n]).abs()
        assert float(dv.mean()) <= 2e-3 * float(g["lr"]) and float(dv.max()) <= 1.01 * float(g["lr"]), (n, float(dv.mean()), float(dv.max()))
    # ---- evaluation at the reference's 2-layer weights after its 60 steps
    if tag == "ckptend":
        load_state()
        model.eval()
        ret = batch_test.test(model, list(data.test_set.keys()), drop_flag=True)
        got = np.concatenate([ret["recall"], ret["ndcg"]])
        want_m = g["eval_metrics"][list(g["eval_steps"]).index(int(g["n_steps"]))]
        assert np.abs(got - want_m).max() <= 1e-4, (got, want_m)


def test_deep_one_call_step_equals_the_launch_by_launch_step(golden, ngcf_data_root):
    """spex_ngcf_deep_step_bce_f32 (L = 2 and 3: whole-table forward of the earlier layers, last layer at the batch's rows, scoring, rows
    backward + push, dense 4-wave layer backwards with their A^T products, Adam — one native call) against NGCFStepper's launch-by-launch
    path (every layer's forward over the whole table) from the same state: four steps, losses and every parameter (the two differ only
    in the order of float atomics; Adam's first steps amplify nothing here because both see the same gradients to rounding)."""
    from spex_amd.trainer import NGCFStepper
    g = golden("ngcf_epinion2_2layer_ckpt")
    rng = np.random.default_rng(17)
    for layers in ("[64,64]", "[64,64,64]"):
        L = layers.count("64")
        res = []
        for native in (True, False):
            gg = dict(seed=g["seed"], drop_seed=g["drop_seed"], mess_dropout=np.asarray([0.1] * L))
            data, model, _ = _ngcf_epinion2_model(gg, ngcf_data_root, layer_size=layers)
            assert model.n_layers == L
            model.train()
            st = NGCFStepper(model, lr=1e-3)
            if not native:
                st._one_call_ok = lambda *a: False
            acc = torch.zeros(1, device=DEV)
            losses = []
            brng = np.random.default_rng(5)
            for k in range(4):
                u = torch.from_numpy(brng.integers(0, data.n_users, 256)).to(DEV)
                i = torch.from_numpy(brng.integers(0, data.n_items, 256)).to(DEV)
                y = torch.from_numpy((brng.random(256) < 1 / 6).astype(np.float32)).to(DEV)
                u[:6] = u[0]                                                         # repeated rows inside the batch
                before = acc.item()
                st.step(u, i, y, loss_acc=acc)
                losses.append(acc.item() - before)
            assert (getattr(st, "_deep_desc", None) is not None) == native and st.t == 4 and model.dropout_step == 4
            res.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
        (l_a, p_a), (l_b, p_b) = res
        assert np.abs(np.asarray(l_a) - np.asarray(l_b)).max() <= 2e-3, (l_a, l_b)        # differences of a running fp32 sum near 1 700: ulp 1.2e-4
        for n in p_a:
            dv = (p_a[n] - p_b[n]).abs()
            assert float(dv.mean()) <= 2e-6 and float(dv.max()) <= 1e-3 + 1e-7, (layers, n, float(dv.mean()), float(dv.max()))


def test_row_sparse_forward_equals_the_whole_table_forward_at_the_batch_rows(epinion2):
    """What the one-call NGCF step runs since round 3: side = A ego at the batch's rows (spex_spmm_rowlist_f32) and the layer at
    those rows (spex_ngcf_layer_fwd_rows_f32, mask indexed by the ROW) against the whole-table launches — bit-identical rows of
    `side` and of the concatenated table (hub rows, repeated users, the isolated pad row, an out-of-range index in the batch),
    every other row of the output untouched."""
    from spex_amd import ops
    from spex_amd.graph import SpexGraph, ngcf_norm_adj
    tr = epinion2["train"]
    rowptr, col, val = ngcf_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    n, n_u = len(rowptr) - 1, 3186
    g = SpexGraph(rowptr, col, val)
    rng = np.random.default_rng(4)
    ego = torch.from_numpy((rng.normal(size=(n, 64)) * 0.1).astype(np.float32)).to(DEV)
    W_gc, W_bi = (torch.from_numpy(rng.normal(size=(64, 64)).astype(np.float32) * 0.2).to(DEV) for _ in range(2))
    b_gc, b_bi = (torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1).to(DEV) for _ in range(2))
    drop = (0.1, 12345, 7)
    side_a = g.spmm(ego)
    out_a = torch.zeros(n, 128, device=DEV)
    ops.ngcf_layer_fwd(ego, side_a, W_gc, b_gc, W_bi, b_bi, out_a, 0, True, drop=drop, pad_row=3185)
    B = 250                                                       # not a multiple of 16: the last tile is partly empty
    deg = np.diff(rowptr)
    users, items = rng.integers(0, 3185, B), rng.integers(0, 12407, B)
    users[:3] = np.argsort(-deg[:3185])[:3]
    items[:3] = np.argsort(-deg[n_u:])[:3]
    users[5:9] = users[0]
    users[9] = 3185                                               # the pad row (isolated: side = 0)
    items[10] = 12407                                             # out of range: skipped
    u_d, i_d = torch.from_numpy(users).to(DEV), torch.from_numpy(items).to(DEV)
    side_b, out_b = torch.full((n, 64), 7.0, device=DEV), torch.full((n, 128), 7.0, device=DEV)
    g.spmm_rows(ego, u_d, i_d, 0, n_u, Y=side_b)
    ops.ngcf_layer_fwd_rows(ego, side_b, W_gc, b_gc, W_bi, b_bi, out_b, u_d, i_d, n_u, drop=drop, pad_row=3185)
    rows = np.unique(np.r_[users, items[items < 12407] + n_u])
    rows_d = torch.from_numpy(rows).to(DEV)
    assert torch.equal(side_b[rows_d], side_a[rows_d])
    assert torch.equal(out_b[rows_d], out_a[rows_d])
    rest = torch.ones(n, dtype=torch.bool, device=DEV)
    rest[rows_d] = False
    assert (out_b[rest] == 7.0).all() and (side_b[rest] == 7.0).all()


def test_fused_scoring_and_rows_backward_equals_the_two_launches():
    """spex_ngcf_score_bwd_rows_f32 against spex_score_bce_slots_f32 + spex_ngcf_layer_bwd_rows_f32 on the same inputs: B = 200
    (the last tile is partial and one tile straddles the user / item boundary of the slots), repeated users, message dropout
    on, one sample with an out-of-range item (skipped by both: zero rows, zero loss)."""
    from spex_amd import ops
    rng = np.random.default_rng(78)
    n, n_u, B = 2000, 700, 200
    ego, side = (torch.from_numpy(rng.normal(size=(n, 64)).astype(np.float32) * 0.3).to(DEV) for _ in range(2))
    W_gc, W_bi = (torch.from_numpy(rng.normal(size=(64, 64)).astype(np.float32) * 0.2).to(DEV) for _ in range(2))
    b_gc, b_bi = (torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1).to(DEV) for _ in range(2))
    all_emb = torch.cat([ego, torch.from_numpy(rng.normal(size=(n, 64)).astype(np.float32) * 0.2).to(DEV)], dim=1).contiguous()
    u_np, i_np = rng.integers(0, n_u, B), rng.integers(0, n - n_u, B)
    u_np[:4] = u_np[10]
    i_np[33] = 5000                                                         # out of range
    users, items = torch.from_numpy(u_np).to(DEV), torch.from_numpy(i_np).to(DEV)
    y = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(DEV)
    drop = (0.1, 99, 3)
    n_parts = ops.ngcf_bwd_rows_parts(2 * B)
    # two launches
    slots = torch.zeros(2 * B, 128, device=DEV)
    loss_a = torch.zeros(1, device=DEV)
    ops.score_bce(all_emb[:n_u], all_emb[n_u:], users, items, y, None, None, 1.0 / B, loss_sum=loss_a, grad_slots=slots, want_gamma=False)
    parts_a = torch.zeros(n_parts, 2 * (64 * 64 + 64), device=DEV)
    gs_a, ge_a = torch.zeros(2 * B, 64, device=DEV), torch.zeros(2 * B, 64, device=DEV)
    ops.ngcf_layer_bwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, slots, 0, None, users, items, n_u, gs_a, ge_a, parts_a, drop=drop, pad_row=n_u)
    # one launch
    per = torch.full((B,), 7.0, device=DEV)
    parts_b = torch.full((n_parts, 2 * (64 * 64 + 64)), 7.0, device=DEV)
    gs_b, ge_b = torch.zeros(2 * B, 64, device=DEV), torch.zeros(2 * B, 64, device=DEV)
    ops.ngcf_score_bwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, all_emb, y, 1.0 / B, users, items, n_u, per, gs_b, ge_b, parts_b,
                            drop=drop, pad_row=n_u)
    assert per[33].item() == 0.0
    assert abs(per.sum().item() - loss_a.item()) <= 1e-5 * abs(loss_a.item())
    assert rel_err(gs_b.cpu().numpy(), gs_a.cpu().numpy()) <= 2e-6
    assert rel_err(ge_b.cpu().numpy(), ge_a.cpu().numpy()) <= 2e-6
    assert (gs_b[33] == 0).all() and (gs_b[B + 33] == 0).all() and (ge_b[33] == 0).all()
    assert rel_err(parts_b.sum(0).cpu().numpy(), parts_a.sum(0).cpu().numpy()) <= 2e-6


@pytest.mark.parametrize("B", [200, 203, 5])
def test_forward_scoring_and_rows_backward_in_one_launch(B):
    """spex_ngcf_fwd_score_bwd_rows_f32 (tiles holding both rows of 8 samples: the layer's forward at the batch's rows, the scores and
    the rows backward, no concatenated table) against spex_ngcf_layer_fwd_rows_f32 + spex_ngcf_score_bwd_rows_f32 on the same inputs:
    batches that are not multiples of 8, repeated users, message dropout on, the pad row, one sample with an out-of-range item
    (zero rows, zero loss)."""
    from spex_amd import ops
    rng = np.random.default_rng(79 + B)
    n, n_u = 2000, 700
    ego, side = (torch.from_numpy(rng.normal(size=(n, 64)).astype(np.float32) * 0.3).to(DEV) for _ in range(2))
    W_gc, W_bi = (torch.from_numpy(rng.normal(size=(64, 64)).astype(np.float32) * 0.2).to(DEV) for _ in range(2))
    b_gc, b_bi = (torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1).to(DEV) for _ in range(2))
    u_np, i_np = rng.integers(0, n_u - 1, B), rng.integers(0, n - n_u, B)
    u_np[:3] = u_np[4]
    i_np[2] = 5000                                                          # out of range
    u_np[1] = n_u - 1                                                       # the pad row (mask numbering skips it)
    users, items = torch.from_numpy(u_np).to(DEV), torch.from_numpy(i_np).to(DEV)
    y = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(DEV)
    drop = (0.1, 99, 3)
    n_parts = ops.ngcf_bwd_rows_parts(2 * B)
    z = lambda *sh: torch.zeros(*sh, device=DEV)
    # two launches: the layer at the batch's rows into the table, then scoring + rows backward from the table
    all_emb = z(n, 128)
    ops.ngcf_layer_fwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, all_emb, users, items, n_u, drop=drop, pad_row=n_u - 1)
    per_a, parts_a, gs_a, ge_a = torch.full((B,), 7.0, device=DEV), z(n_parts, 2 * (64 * 64 + 64)), z(2 * B, 64), z(2 * B, 64)
    ops.ngcf_score_bwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, all_emb, y, 1.0 / B, users, items, n_u, per_a, gs_a, ge_a, parts_a,
                            drop=drop, pad_row=n_u - 1)
    # one launch
    per_b, parts_b, gs_b, ge_b = torch.full((B,), 7.0, device=DEV), torch.full((n_parts, 2 * (64 * 64 + 64)), 7.0, device=DEV), z(2 * B, 64), z(2 * B, 64)
    ops.ngcf_fwd_score_bwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, y, 1.0 / B, users, items, n_u, per_b, gs_b, ge_b, parts_b,
                                drop=drop, pad_row=n_u - 1)
    assert per_b[2].item() == 0.0 and per_a[2].item() == 0.0
    assert rel_err(per_b.cpu().numpy(), per_a.cpu().numpy()) <= 2e-6
    assert rel_err(gs_b.cpu().numpy(), gs_a.cpu().numpy()) <= 3e-6
    assert rel_err(ge_b.cpu().numpy(), ge_a.cpu().numpy()) <= 3e-6
    assert (gs_b[2] == 0).all() and (gs_b[B + 2] == 0).all() and (ge_b[2] == 0).all()
    assert rel_err(parts_b.sum(0).cpu().numpy(), parts_a.sum(0).cpu().numpy()) <= 3e-6


def test_layer_backward_rows_form_equals_dense_form(oracle):
    """spex_ngcf_layer_bwd_rows_f32 (compact tiles over the batch's slots, every slot with its own gradient row) against
    the dense form fed with the same gradients scattered into a table: the layer's backward is linear in the upstream
    gradient, so a row's dense result equals the sum of its slots' compact results; the weight gradients (summed from the
    partial blocks, also through spex_adam_step_sum_f32) are the same."""
    from spex_amd import ops
    rng = np.random.default_rng(77)
    n = 2000
    ego, side = (torch.from_numpy(rng.normal(size=(n, 64)).astype(np.float32) * 0.3).to(DEV) for _ in range(2))
    W_gc, W_bi = (torch.from_numpy(rng.normal(size=(64, 64)).astype(np.float32) * 0.2).to(DEV) for _ in range(2))
    b_gc, b_bi = (torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1).to(DEV) for _ in range(2))
    u_np, i_np = rng.integers(0, 700, 200), rng.integers(0, 1200, 200)
    u_np[:4] = u_np[10]                                                     # a user named by five slots
    users, items = torch.from_numpy(u_np).to(DEV), torch.from_numpy(i_np).to(DEV)
    slot_rows = torch.from_numpy(np.concatenate([u_np, i_np + 700])).to(DEV)
    g_slots = torch.from_numpy(rng.normal(size=(400, 128)).astype(np.float32)).to(DEV)
    g_all = torch.zeros(n, 128, device=DEV).index_add_(0, slot_rows, g_slots)
    drop = (0.1, 99, 3)
    gW = [torch.zeros(64, 64, device=DEV), torch.zeros(64, device=DEV), torch.zeros(64, 64, device=DEV), torch.zeros(64, device=DEV)]
    gs, ge = torch.empty(n, 64, device=DEV), torch.empty(n, 64, device=DEV)
    ops.ngcf_layer_bwd(ego, side, W_gc, b_gc, W_bi, b_bi, g_all, 0, None, gs, ge, *gW, drop=drop, pad_row=700)
    n_parts = ops.ngcf_bwd_rows_parts(400)
    assert n_parts == 25
    parts = torch.full((n_parts, 2 * (64 * 64 + 64)), 7.0, device=DEV)
    gsc, gec = torch.full((400, 64), 9.0, device=DEV), torch.full((400, 64), 9.0, device=DEV)
    ops.ngcf_layer_bwd_rows(ego, side, W_gc, b_gc, W_bi, b_bi, g_slots, 0, None, users, items, 700, gsc, gec, parts, drop=drop, pad_row=700)
    got_s = torch.zeros(n, 64, device=DEV).index_add_(0, slot_rows, gsc)
    got_e = torch.zeros(n, 64, device=DEV).index_add_(0, slot_rows, gec)
    assert rel_err(got_s.cpu().numpy(), gs.cpu().numpy()) <= 2e-6
    assert rel_err(got_e.cpu().numpy(), ge.cpu().numpy()) <= 2e-6
    tot = parts.sum(0).cpu().numpy()
    want = np.concatenate([gW[0].cpu().numpy().ravel(), gW[1].cpu().numpy(), gW[2].cpu().numpy().ravel(), gW[3].cpu().numpy()])
    assert rel_err(tot, want) <= 1e-5
    # the optimiser pass that sums the blocks == plain Adam on the summed gradient
    p1 = torch.from_numpy(rng.normal(size=len(want)).astype(np.float32)).to(DEV)
    p2, m1, v1, m2, v2 = p1.clone(), torch.zeros_like(p1), torch.zeros_like(p1), torch.zeros_like(p1), torch.zeros_like(p1)
    ops.adam_step_sum(p1, parts, m1, v1, 1, lr=1e-2)
    ops.adam_step(p2, parts.sum(0).contiguous(), m2, v2, 1, lr=1e-2)
    assert rel_err(p1.cpu().numpy(), p2.cpu().numpy()) <= 1e-6
    # and the scoring kernel's per-sample rows are what sums up to its table form
    tab = torch.from_numpy(rng.normal(size=(n, 128)).astype(np.float32)).to(DEV)
    y = torch.from_numpy((rng.random(200) < 0.3).astype(np.float32)).to(DEV)
    gt_tab, slots = torch.zeros(n, 128, device=DEV), torch.zeros(400, 128, device=DEV)
    ops.score_bce(tab[:700], tab[700:], users, items, y, gt_tab[:700], gt_tab[700:], 1.0 / 200, grad_slots=slots, want_gamma=False)
    assert rel_err(torch.zeros(n, 128, device=DEV).index_add_(0, slot_rows, slots).cpu().numpy(), gt_tab.cpu().numpy()) <= 2e-6
