"""SPEX_STEP_DETERMINISTIC — the atomic-free accumulation mode of the three one-call training steps.

The fast path adds the batch's gradient rows with float atomics (duplicate users / items of a batch, the push-form first
backward product), so two runs from one state differ in the last bits and Adam amplifies that over thousands of steps
(round 2's NGCF whole-epoch gates had to absorb it).  In the deterministic mode every sum has a fixed order: per-sample
rows added per table row in ascending slot order (spex_reduce_slots_f32 — the order of the reference's CPU index backward),
the backward propagation in pull form.  Tested here: the reduce kernel bit for bit against sequential NumPy adds, and for
every stepper that two runs of 300 steps from the same state end in BIT-IDENTICAL parameters (run once each — the test is the
equality, not a flake count), that the mode agrees with the fast path to rounding, and with the oracle on one step."""
import argparse
import os
import random

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def training_batches(epinion2, n_steps, B=256, seed=5):
    """Batches shaped like the reference's: a random observed pair or one of its five same-user negatives (users and
    positive items arrive in proportion to their degree, so users repeat inside a batch all the time)."""
    rng = np.random.default_rng(seed)
    train = epinion2["train"]
    out = []
    for _ in range(n_steps):
        k = rng.integers(0, len(train), B)
        u = train[k, 0].copy()
        i = train[k, 1].copy()
        neg = rng.random(B) < 5 / 6
        i[neg] = rng.integers(0, 12407, int(neg.sum()))
        out.append((u.astype(np.int64), i.astype(np.int64), (~neg).astype(np.float32)))
    return out


# ---------------------------------------------------------------------------------------------- the reduce kernel
def test_reduce_slots_adds_in_ascending_slot_order():
    """spex_reduce_slots_f32 against sequential fp32 adds in slot order (np.add.at walks the index list in order): bit-identical
    — with heavy repetition (one row named 200 times), an out-of-range slot, the accumulate form and the clear form."""
    from spex_amd import ops
    rng = np.random.default_rng(3)
    n_rows, n_u = 5000, 2000
    ua, ub = rng.integers(0, n_u, 300), rng.integers(0, n_rows - n_u, 300)
    ua[:200:1] = np.where(rng.random(200) < 0.7, 17, ua[:200])        # row 17 many times
    ub[5:40] = 123
    ua[250] = 10 ** 9                                                 # out of range: skipped
    slots = (rng.normal(size=(600, 64)) * np.exp(rng.normal(size=(600, 1)) * 3)).astype(np.float32)   # wide dynamic range
    rows = np.concatenate([ua, ub + n_u])
    ok = (rows >= 0) & (rows < n_rows)
    want = np.zeros((n_rows, 64), np.float32)
    np.add.at(want, rows[ok], slots[ok])
    out = torch.full((n_rows, 64), 7.0, device=DEV)
    ops.reduce_slots(t(ua), t(ub), n_u, n_rows, t(slots), out)
    got = out.cpu().numpy()
    touched = np.zeros(n_rows, bool); touched[rows[ok]] = True
    assert np.array_equal(got[touched], want[touched])                # bit for bit
    assert (got[~touched] == 7.0).all()                               # rows no slot names are left alone
    # accumulate form: out += the same sums; scale
    base = rng.normal(size=(n_rows, 64)).astype(np.float32)
    out2 = t(base.copy())
    ops.reduce_slots(t(ua), t(ub), n_u, n_rows, t(slots), out2, scale=0.25, accumulate=True)
    want2 = base.copy()
    want2[touched] = base[touched] + want[touched] * np.float32(0.25)
    assert np.array_equal(out2.cpu().numpy(), want2)
    # clear form
    ops.reduce_slots(t(ua), t(ub), n_u, n_rows, None, out)
    got = out.cpu().numpy()
    assert (got[touched] == 0.0).all() and (got[~touched] == 7.0).all()
    # a strided slot array (NGCF's [2B, 128] per-sample rows)
    wide = np.zeros((600, 128), np.float32); wide[:, :64] = slots
    out3 = torch.zeros(n_rows, 64, device=DEV)
    ops.reduce_slots(t(ua), t(ub), n_u, n_rows, t(wide), out3)
    assert np.array_equal(out3.cpu().numpy()[touched], want[touched])


# ---------------------------------------------------------------------------------------------- LightGCN
def _lightgcn_stepper(golden, epinion2, deterministic, G):
    from spex_amd.datasets import epinion2_tables
    from spex_amd.graph import lightgcn_norm_adj
    from spex_amd.trainer import LightGCNStepper
    csr = lightgcn_norm_adj(epinion2["train"][:, 0], epinion2["train"][:, 1], 3185, 12407)
    uw, iw = epinion2_tables(3186, 12407)
    E0 = t(np.concatenate([uw, iw]))
    return LightGCNStepper(G(*csr), E0, 3186, n_layers=3, lr=1e-3, deterministic=deterministic), csr


@pytest.fixture(scope="module")
def G():
    from spex_amd.graph import SpexGraph
    return lambda *a, **k: SpexGraph(*a, **k)


def test_lightgcn_deterministic_step_repeats_bit_for_bit(golden, epinion2, oracle, G):
    """Two 300-step runs of the deterministic one-call step on Epinion2 from the same table, on training-shaped batches (users
    repeat inside every batch): BIT-IDENTICAL tables, moments and loss sums.  The launch-by-launch form of the same mode
    (score_bce slots -> reduce_slots -> pull-form backward) equals the one-call form bit for bit; the fast path (float
    atomics + push form) agrees to rounding; one deterministic step equals the oracle's step."""
    batches = training_batches(epinion2, 300)
    runs = []
    for rep in range(2):
        st, csr = _lightgcn_stepper(golden, epinion2, True, G)
        acc = torch.zeros(1, device=DEV)
        for u, i, y in batches:
            st.step_bce(t(u), t(i), t(y), loss_acc=acc, batch_rows_only=True)
        runs.append((st.E0.clone(), st.m.clone(), st.v.clone(), acc.clone()))
        assert st.t == 300
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    # launch by launch, same mode: same bits
    st, _ = _lightgcn_stepper(golden, epinion2, True, G)
    acc = torch.zeros(1, device=DEV)
    for u, i, y in batches[:20]:
        st.step_bce(t(u), t(i), t(y), loss_acc=acc, batch_rows_only=False)
    st2, _ = _lightgcn_stepper(golden, epinion2, True, G)
    acc2 = torch.zeros(1, device=DEV)
    for u, i, y in batches[:20]:
        st2.step_bce(t(u), t(i), t(y), loss_acc=acc2, batch_rows_only=True)
    assert torch.equal(st.E0, st2.E0)
    # the fast path: same function, sums associated differently
    fast, _ = _lightgcn_stepper(golden, epinion2, False, G)
    accf = torch.zeros(1, device=DEV)
    for u, i, y in batches:
        fast.step_bce(t(u), t(i), t(y), loss_acc=accf, batch_rows_only=True)
    assert abs(accf.item() - runs[0][3].item()) <= 2e-5 * abs(accf.item())
    assert rel_err(fast.E0.cpu().numpy(), runs[0][0].cpu().numpy()) <= 2e-4       # 300 Adam steps of +-lr on rounding noise
    # one step against the oracle (pull form, ascending column order, index-order duplicate sums)
    st, csr = _lightgcn_stepper(golden, epinion2, True, G)
    E0 = st.E0.cpu().numpy().copy()
    u, i, y = batches[0]
    acc = torch.zeros(1, device=DEV)
    st.step_bce(t(u), t(i), t(y), loss_acc=acc, batch_rows_only=True)
    _, loss_o, Gd = oracle.lightgcn_loss_and_grad(*csr, E0, 3186, 3, u, i, y)
    W = E0.copy()
    oracle.adam_step(W, Gd, np.zeros_like(W), np.zeros_like(W), 1)
    assert abs(acc.item() / 256 - float(loss_o)) <= 2e-6
    assert rel_err(st.E0.cpu().numpy(), W) <= 5e-6


# ---------------------------------------------------------------------------------------------- NGCF
def test_ngcf_deterministic_step_repeats_bit_for_bit(epinion2):
    """Two 300-step runs of the deterministic NGCF step (message dropout on) on Epinion2 from the same seed: bit-identical
    tables, layer weights and loss sums; the fast path agrees in its losses (its parameters walk apart along NGCF's
    scale-invariant direction — the reason this mode exists)."""
    from spex_amd.graph import ngcf_norm_adj
    from spex_amd.ngcf import NGCF
    from spex_amd.trainer import NGCFStepper
    import scipy.sparse as sp
    tr = epinion2["train"]
    n_users, n_items = 3185, 12407
    rowptr, col, val = ngcf_norm_adj(tr[:, 0], tr[:, 1], n_users, n_items)
    adj = sp.csr_matrix((val, col, rowptr), shape=(n_users + n_items, n_users + n_items))
    args = argparse.Namespace(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")
    batches = training_batches(epinion2, 300, seed=9)

    def run(det):
        torch.manual_seed(2020)
        model = NGCF({"n_users": n_users, "n_items": n_items, "norm_adj": adj}, DEV, args).to(DEV)
        model.message_dropout_seed = 2020
        model.train()
        st = NGCFStepper(model, lr=1e-3, deterministic=det)
        acc = torch.zeros(1, device=DEV)
        for u, i, y in batches:
            st.step(t(u), t(i), t(y), loss_acc=acc)
        assert st.t == 300 and model.dropout_step == 300
        return st.E0.clone(), st.W.clone(), st.mE.clone(), acc.clone()
    a, b = run(True), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    f = run(False)
    assert abs(f[3].item() - a[3].item()) <= 5e-5 * abs(a[3].item())
    assert rel_err(f[0].cpu().numpy(), a[0].cpu().numpy()) <= 5e-3


# ---------------------------------------------------------------------------------------------- dual task
from test_gpu_dropin import data_root, _dual_epinion2, _epinion2_trust_raw   # noqa: E402,F401  (fixture + builders)


def test_dual_task_deterministic_step_repeats_bit_for_bit(data_root, golden):
    """Two 120-step runs of the deterministic dual-task step (rec branch + trust head + uncertainty weights, two streams) on
    Epinion2 with the reference's own trust paths: every parameter of the arena, both Adam moments and both loss sums are
    bit-identical; the fast path agrees to rounding in its losses and task weights."""
    from collections import defaultdict
    import utility1.dataloader as dl
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper, train_epoch_dual
    g = golden("dual_epinion2_epochs")
    raw_train, _ = _epinion2_trust_raw(golden)
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    cap = 3 * int(g["trust_batch_size"])

    def run(det, pipelined=True):
        os.environ["SPEX_DUAL_PIPELINED"] = "1" if pipelined else "0"
        args, dataset, net = _dual_epinion2(data_root)                  # includes set_seed: same negatives, shuffle, path cuts
        td = dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat)
        train2 = Data(raw_train, dataset.n_users, shuffle=False)
        net = net.to(DEV)
        st = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr, deterministic=det)
        totals = train_epoch_dual(st, td, train2, by_user, cap, max_steps=120)
        torch.cuda.synchronize()
        return st.arena.clone(), st.m.clone(), st.v.clone(), totals.clone()
    try:
        a, b = run(True), run(True)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        # the pipelined form (Adam split by owner over the two streams, what train_epoch_dual runs by default) does the same
        # arithmetic as the fork / join form: bit-identical in the deterministic mode
        c = run(True, pipelined=False)
        for x, y in zip(a, c):
            assert torch.equal(x, y)
        f = run(False)
    finally:
        os.environ.pop("SPEX_DUAL_PIPELINED", None)
    assert rel_err(f[3].cpu().numpy(), a[3].cpu().numpy()) <= 5e-5
    assert (f[0][-2:] - a[0][-2:]).abs().max().item() <= 2e-5            # the task weights


def test_dual_task_pipelined_steps_equal_forked_steps(data_root, golden):
    """DualTaskStepper.step driven directly, 40 deterministic steps with 0..15 paths per step (steps WITHOUT paths included: the
    side stream then only carries its share of the Adam pass), a join + an outside read in the middle: the pipelined form, the
    fork / join form and the one-stream form end in bit-identical arenas, moments and loss sums."""
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper
    raw_train, _ = _epinion2_trust_raw(golden)
    rng = np.random.default_rng(5)
    B, steps = 256, 40
    n_paths = [int(rng.integers(0, 16)) for _ in range(steps)]
    n_paths[3] = n_paths[17] = 0

    def run(mode):
        args, dataset, net = _dual_epinion2(data_root)
        train2 = Data(raw_train, dataset.n_users, shuffle=False)
        net = net.to(DEV)
        st = DualTaskStepper(net, path_capacity=15, path_len=train2.len_max, lr=args.lr, deterministic=True,
                             two_streams=mode != "one", pipelined=mode == "pipelined")
        r = np.random.default_rng(6)
        users = torch.from_numpy(r.integers(0, dataset.n_users, (steps, B))).to(DEV)
        items = torch.from_numpy(r.integers(0, dataset.m_items, (steps, B))).to(DEV)
        labels = torch.from_numpy((r.random((steps, B)) < 1 / 6).astype(np.float32)).to(DEV)
        picks = [r.integers(0, len(raw_train[0]), k) for k in n_paths]
        staged = []
        for pk in picks:
            if len(pk) == 0:
                staged.append((None, None, None))
                continue
            inputs, mask, targets = train2.get_slice(pk)
            staged.append((torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.int64)).to(DEV),
                           torch.from_numpy(np.asarray(mask).sum(1).astype(np.int64)).to(DEV),
                           torch.from_numpy(np.asarray(targets).astype(np.int64)).to(DEV)))
        torch.cuda.synchronize()
        mid = None
        for k in range(steps):
            st.step(users[k], items[k], labels[k], *staged[k])
            if k == 20:
                st.join()
                mid = st.loss_acc.clone()                 # an outside read on the current stream, after the join
        st.join()
        torch.cuda.synchronize()
        return st.arena.clone(), st.m.clone(), st.v.clone(), st.loss_acc.clone(), mid
    a, b, c = run("pipelined"), run("forked"), run("one")
    for other in (b, c):
        for x, y in zip(a, other):
            assert torch.equal(x, y)
    assert torch.isfinite(a[0]).all() and a[3][0].item() > 0 and a[3][1].item() > 0
