"""BASELINE config 5 (main_auto_expert_s.py dual-task on Weibo) and config 4 (NGCF on Twitter) ON THEIR OWN SHAPES.

The reference ships neither dataset (README.md:81: Google Drive only), so the workload is synthetic with the published sizes —
6 812 Weibo users / 8 930 Twitter users (Trust_SPEX/code/main_trust.py:42,44), 20 000 items, ~400 k interactions with log-normal
user activity and Zipf item popularity (hub rows far beyond 1 024 stored entries, empty rows), synthetic trust paths of 1..6
users — and the reference cannot be run on it here.  What pins these tests instead: the fused kernels against their own
multi-launch sequences, the one-call steps against the drop-in model under autograd + torch Adam, and both against
oracle/trust_oracle.py's fp64 restatement of the whole dual-task forward (itself pinned to the reference's G11 golden in the CPU
suite).  Round 2 ran the dual-task kernels on Epinion2 only (3 185 users, longest row 1 020 entries): the gated batch kernel's
segment dealing had never seen a row of more than 1 024 entries, nor the NGCF step's push over such rows."""
import argparse
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
N_USERS, N_ITEMS, N_EDGES = 6812, 20000, 400000


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def weibo_files(root):
    """The synthetic Weibo-shaped interactions as the reference's rating files + synthetic trust paths (lists as
    main_auto_expert_s.py:38-41 unpickles them)."""
    from spex_amd.datasets import materialise_rating_files, synthetic_interactions
    u, i = synthetic_interactions(N_USERS, N_ITEMS, N_EDGES, seed=7, sigma=1.4)      # user hubs of ~1 800 interactions
    train = np.stack([u.numpy(), i.numpy()], 1)
    rng = np.random.default_rng(17)
    test_users = np.arange(N_USERS)
    test_pos = rng.integers(0, N_ITEMS, N_USERS)
    test_neg = rng.integers(0, N_ITEMS, (N_USERS, 99))
    data_root = materialise_rating_files(root, "weibo", train, test_users, test_pos, test_neg)
    # trust paths: 1..6 users each, heavy-tailed start users (so that a batch's users bring many paths), a target user
    w = np.exp(rng.normal(size=N_USERS))
    starts = rng.choice(N_USERS, 30000, p=w / w.sum())
    paths = [[int(s)] + rng.integers(0, N_USERS, int(rng.integers(0, 6))).tolist() for s in starts]
    targets = rng.integers(0, N_USERS, len(paths)).tolist()
    return data_root, train, (paths, targets)


@pytest.fixture(scope="module")
def weibo(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("weibo"))
    data_root, train, raw_train = weibo_files(root)
    return dict(data_root=data_root, train=train, raw_train=raw_train)


def build_dual(data_root):
    import lg_parser
    import utility1.dataloader as dataloader
    import utility1.model_expert_s as mex
    import utility1.utils as utils
    args = lg_parser.parse_args_r(["--dataset", "weibo", "--data_path", data_root])
    utils.set_seed(args.seed)
    dataset = dataloader.Loader(args)
    return args, dataset, mex.LightGCN(args, dataset)


def hub_batch(csr, n_u, B, rng):
    """A batch whose rows include the heaviest user and item rows (> 1 024 stored entries), an empty row if there is one,
    repeats, and uniformly drawn others."""
    deg = np.diff(csr[0])
    users, items = rng.integers(0, n_u - 1, B), rng.integers(0, len(deg) - n_u, B)
    users[:6] = np.argsort(-deg[: n_u - 1])[:6]
    items[:6] = np.argsort(-deg[n_u:])[:6]
    users[10:14] = users[0]
    items[20:24] = items[0]
    empty = np.flatnonzero(deg[n_u:] == 0)
    if len(empty):
        items[30] = empty[0]
    labels = (rng.random(B) < 1 / 6).astype(np.float32)
    return users.astype(np.int64), items.astype(np.int64), labels


def test_gated_batch_forward_with_hub_rows_beyond_1024_entries(weibo):
    """spex_gated_batch_fwd_f32 on the Weibo-shaped graph with the heaviest rows in the batch (users of > 1 024 interactions,
    items of > 1 024 customers: 16 virtual waves chaining several 64-entry segments each) against the three launches it
    replaces — spmm_rows -> expert_gate_rows -> score_bce(grad_slots): the propagated rows bit-identical (same segments, same
    order), loss and gradient rows to rounding."""
    from spex_amd import ops
    args, dataset, net = build_dual(weibo["data_root"])
    csr = dataset.build_adjacency()
    deg = np.diff(csr[0])
    n, n_u, L, B = len(deg), N_USERS + 1, 3, 256
    assert deg[:n_u].max() > 1024 and deg[n_u:].max() > 1024
    g = net.Graph if not isinstance(net.Graph, (list, tuple)) else None
    from spex_amd.graph import SpexGraph
    g = SpexGraph(*csr, device=DEV)
    rng = np.random.default_rng(21)
    X, run, raw = (t((rng.normal(size=(n, 64)) * s).astype(np.float32)) for s in (0.1, 0.05, 0.1))
    att_u, att_i = t((rng.normal(size=(128, 2)) * 0.5).astype(np.float32)), t((rng.normal(size=(128, 2)) * 0.5).astype(np.float32))
    users, items, labels = hub_batch(csr, n_u, B, rng)
    assert deg[users[0]] > 1024 and deg[items[0] + n_u] > 1024
    u_d, i_d, y_d = t(users), t(items), t(labels)
    lo_a = torch.zeros(n, 64, device=DEV)
    g.spmm_rows(X, u_d, i_d, 0, n_u, acc_in=run, acc_out=lo_a, acc_div=float(L + 1))
    mixed = ops.expert_gate_rows(raw, lo_a, att_u, att_i, u_d, i_d, n_u)
    slots_a, loss_a = torch.zeros(2 * B, 64, device=DEV), torch.zeros(1, device=DEV)
    ar = torch.arange(B, device=DEV)
    ops.score_bce(mixed[:B], mixed[B:], ar, ar, y_d, None, None, 1.0 / B, loss_sum=loss_a, grad_slots=slots_a, want_gamma=False)
    lo_b, slots_b, loss_b = torch.zeros(n, 64, device=DEV), torch.full((2 * B, 64), 7.0, device=DEV), torch.zeros(1, device=DEV)
    ops.gated_batch_fwd(g, X, run, float(L + 1), raw, att_u, att_i, u_d, i_d, y_d, n_u, 1.0 / B, loss_b, lo_b, slots_b)
    rows = torch.cat([u_d, i_d + n_u])
    assert torch.equal(lo_b[rows], lo_a[rows])
    assert abs(loss_a.item() - loss_b.item()) <= 1e-5 * abs(loss_a.item())
    assert rel_err(slots_b.cpu().numpy(), slots_a.cpu().numpy()) <= 2e-6
    # per-sample losses (the deterministic step's form) add up to the same sum
    per = torch.zeros(B, device=DEV)
    ops.gated_batch_fwd(g, X, run, float(L + 1), raw, att_u, att_i, u_d, i_d, y_d, n_u, 1.0 / B, None, lo_b, slots_b, loss_per_sample=per)
    assert abs(per.double().sum().item() - loss_a.item()) <= 1e-5 * abs(loss_a.item())
    # the propagated rows against the oracle's pull-form rows (hub rows re-associate their segment sums: 1e-5)
    from oracle import oracle as O
    want = (run.cpu().numpy() + O.spmm(*csr, X.cpu().numpy(), n_threads=8)) / np.float32(L + 1)
    assert rel_err(lo_b[rows].cpu().numpy(), want[rows.cpu().numpy()]) <= 1e-5
    # the whole middle in one launch (spex_gated_batch_f32: + the gate's backward + the push) against forward -> gate backward -> push
    z = lambda *sh: torch.zeros(*sh, device=DEV)
    g_prop, g_raw, G_a, ga_u, ga_i = z(n, 64), z(n, 64), z(n, 64), z(128, 2), z(128, 2)
    d_prop = ops.expert_gate_rows_bwd(raw, lo_b, att_u, att_i, u_d, i_d, n_u, slots_b, g_prop, g_raw, ga_u, ga_i)
    ops.spmm_push_batch(g, u_d, i_d, n_u, d_prop, G_a, add=d_prop, scale=1.0 / (L + 1))
    got = (z(1), z(n, 64), z(n, 64), z(n, 64), z(16, 2, 128, 2))
    ops.gated_batch(g, X, run, float(L + 1), raw, att_u, att_i, u_d, i_d, y_d, n_u, 1.0 / B, 1.0 / (L + 1), *got)
    assert abs(got[0].item() - loss_a.item()) <= 1e-5 * abs(loss_a.item())
    g_att = got[4].sum(0)
    for nm, a_, b_ in zip(("g_prop", "G", "g_raw", "g_att_u", "g_att_i"), (g_prop, G_a, g_raw, ga_u, ga_i), got[1:4] + (g_att[0], g_att[1])):
        assert rel_err(b_.cpu().numpy(), a_.cpu().numpy()) <= 5e-6, nm


def _paths_for(users, by_user, cap, rng):
    chosen = []
    for u in sorted(set(users.tolist())):
        chosen.extend(by_user.get(int(u), []))
    if len(chosen) > cap:
        chosen = sorted(rng.choice(chosen, cap, replace=False).tolist())
    return np.asarray(chosen, dtype=int)


@pytest.mark.parametrize("deterministic", [False, True])
def test_dual_task_step_on_the_weibo_shape_vs_autograd_and_the_oracle(weibo, deterministic):
    """spex_dual_task_step_f32 (DualTaskStepper) on the Weibo-shaped workload — batches that name rows of > 1 024 stored entries
    (the rec branch's push / pull products and the gated batch kernel on hubs), 15 synthetic trust paths per step over a
    6 812-user table (the logits sweep's real width) — three steps against (a) the drop-in model under autograd + torch Adam:
    both losses every step, every parameter afterwards; (b) oracle/trust_oracle.py's fp64 restatement of the whole forward
    (pinned to the reference's G11): both losses of the first step, and the autograd path's gradients of loss1 + loss2."""
    from collections import defaultdict
    from oracle.trust_oracle import dual_task_losses
    from utility2.utils import Data
    from spex_amd.trainer import DualTaskStepper
    args, dataset, net = build_dual(weibo["data_root"])
    _, _, ref = build_dual(weibo["data_root"])                       # the same seed: the same initial parameters
    net, ref = net.to(DEV), ref.to(DEV)
    csr = dataset.build_adjacency()
    n_u = N_USERS + 1
    raw_train = weibo["raw_train"]
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2 = Data(raw_train, dataset.n_users, shuffle=False)
    rng = np.random.default_rng(5)
    cap = 15
    st = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr, deterministic=deterministic)
    opt = torch.optim.Adam(ref.parameters(), lr=args.lr)
    ref.train()
    for step in range(3):
        users, items, labels = hub_batch(csr, n_u, 256, rng)
        sl = _paths_for(users, by_user, cap, rng)
        assert len(sl) == cap
        inputs, mask, targets = train2.get_slice(sl)
        if step == 0:                                               # (b) the fp64 restatement at the initial parameters
            P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in ref.state_dict().items() if k != "task_weights"}
            o1, o2 = dual_task_losses(*csr, P, users, items, labels.astype(np.float64), inputs, mask, targets,
                                      nonhybrid=bool(args.nonhybrid))
            (o1 + o2).backward()
        opt.zero_grad()
        l1, l2 = ref(t(users), t(items), t(labels), sl, train2, flag=0)
        w = ref.task_weights
        if step == 0:
            (l1 + l2).backward(retain_graph=True)
            assert abs(l1.item() - o1.item()) <= 2e-6 and abs(l2.item() - o2.item()) <= 2e-5 * o2.item(), (l1.item(), o1.item(), l2.item(), o2.item())
            for name, p in ref.named_parameters():
                if name == "task_weights":
                    continue
                want = P[name].grad.numpy().reshape(p.shape)
                got = p.grad.cpu().numpy()
                err = np.abs(got - want).max()
                assert err <= 5e-5 * max(np.abs(want).max(), 1e-7) or err <= 2e-9, (name, err, np.abs(want).max())
            opt.zero_grad()
        (torch.exp(-2 * w[0]) * l1 + torch.exp(-2 * w[1]) * l2 + 2 * 6 * len(users) * w[0] + len(sl) * w[1]).backward()
        opt.step()
        seq, seq_l, tgt = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(DEV) for a in (inputs, np.asarray(mask).sum(1), targets))
        st.loss_acc.zero_()
        st.step(t(users), t(items), t(labels), seq, seq_l, tgt)
        got = st.loss_acc.cpu().numpy()
        assert abs(got[0] - l1.item()) <= 3e-6 and abs(got[1] - l2.item()) <= 2e-5 * l2.item(), (step, got, l1.item(), l2.item())
        if step == 0:
            assert abs(got[0] - o1.item()) <= 3e-6 and abs(got[1] - o2.item()) <= 2e-5 * o2.item()
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        # Adam's first steps move every touched parameter by ~lr whatever the gradient's size: compare on that scale
        assert (p.detach() - q.detach()).abs().max().item() <= 0.02 * args.lr * 3, name
    assert st.t == 3


def test_ngcf_step_on_the_twitter_shape_vs_autograd():
    """spex_ngcf_step_bce_f32 (NGCFStepper, fast and deterministic) on the Twitter-shaped graph (8 930 users, D^-1 (A + I) with
    hub rows beyond 1 024 entries, non-symmetric: the push walks the rows of A, the pull-form product those of A^T) against the
    drop-in NGCF model under autograd + torch Adam: three steps on batches that name the hubs — losses and every parameter."""
    import scipy.sparse as sp
    from spex_amd.datasets import synthetic_interactions
    from spex_amd.graph import ngcf_norm_adj
    from spex_amd.ngcf import NGCF
    from spex_amd.trainer import NGCFStepper
    n_users, n_items = 8930, 20000
    u, i = synthetic_interactions(n_users, n_items, 400000, seed=7)
    rowptr, col, val = ngcf_norm_adj(u.numpy(), i.numpy(), n_users, n_items)
    deg = np.diff(rowptr)
    assert deg.max() > 1024
    adj = sp.csr_matrix((val, col, rowptr), shape=(n_users + n_items, n_users + n_items))
    args = argparse.Namespace(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")
    rng = np.random.default_rng(8)
    batches = []
    for _ in range(3):
        users, items = rng.integers(0, n_users, 256), rng.integers(0, n_items, 256)
        users[:4] = np.argsort(-deg[:n_users])[:4]
        items[:4] = np.argsort(-deg[n_users:])[:4]
        users[8:12] = users[0]
        batches.append((users.astype(np.int64), items.astype(np.int64), (rng.random(256) < 1 / 6).astype(np.float32)))

    def model():
        torch.manual_seed(2020)
        m = NGCF({"n_users": n_users, "n_items": n_items, "norm_adj": adj}, DEV, args).to(DEV)
        m.message_dropout_seed = 77
        m.train()
        return m
    ref = model()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    want_losses = []
    for users, items, labels in batches:
        opt.zero_grad()
        loss = ref(user=t(users), item=t(items), labels_list=t(labels), flag=0)
        loss.backward()
        opt.step()
        want_losses.append(loss.item())
    for det in (False, True):
        m = model()
        st = NGCFStepper(m, lr=1e-3, deterministic=det)
        for k, (users, items, labels) in enumerate(batches):
            acc = torch.zeros(1, device=DEV)
            st.step(t(users), t(items), t(labels), loss_acc=acc)
            assert abs(acc.item() / 256 - want_losses[k]) <= 2e-5, (det, k, acc.item() / 256, want_losses[k])
        for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            assert (p.detach() - q.detach()).abs().max().item() <= 0.02 * 1e-3 * 3, (det, name)


# ---------------------------------------------------------------------------------------------- two ranks, row-partitioned
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partitioned_worker(rank, world, port, out_dir, root, n_steps):
    import sys
    from collections import defaultdict
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "spex_amd", "dropin"))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from utility2.utils import Data
    from spex_amd.dist_dual import PartitionedDualTask
    import test_gpu_config5_weibo as T
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    data_root, train, raw_train = T.weibo_files(root + f"/r{rank}")
    args, dataset, core = T.build_dual(data_root)
    csr = dataset.build_adjacency()
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2 = Data(raw_train, dataset.n_users, shuffle=False)
    core = core.to(dev)
    model = PartitionedDualTask(core, csr, rank, world, dev)
    opt = torch.optim.Adam(model.trained_parameters(), lr=args.lr)
    core.train()
    rng = np.random.default_rng(5)
    l1s, l2s = [], []
    for step in range(n_steps):
        users, items, labels = T.hub_batch(csr, T.N_USERS + 1, 256, rng)
        sl = T._paths_for(users, by_user, 15, rng)
        opt.zero_grad()
        l1, l2 = model(torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(labels), sl, train2)
        w = model.task_weights
        (torch.exp(-2 * w[0]) * l1 + torch.exp(-2 * w[1]) * l2 + 2 * 6 * len(users) * w[0] + len(sl) * w[1]).backward()
        model.reduce_gate_gradients()
        opt.step()
        l1s.append(l1.item()); l2s.append(l2.item())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss1=np.asarray(l1s), loss2=np.asarray(l2s), r0=model.P.r0, r1=model.P.r1,
             table=model.E0_local.detach().cpu().numpy(), task_weights=model.task_weights.detach().cpu().numpy(),
             w=core.w.detach().cpu().numpy(), hubs=model.P.graph.n_long_rows)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_partitioned_dual_task_on_the_weibo_shape(weibo, tmp_path):
    """BASELINE config 5's multi-GPU form on its own shape: spex_amd.dist_dual.PartitionedDualTask with two ranks (sharing the test
    box's one GPU, collectives through gloo) on the Weibo-shaped graph — hub rows cut across shards, the trust head on the
    all-gathered user block of 6 813 rows — four training steps: both ranks' per-step losses equal the single-device drop-in
    model's (same seeds, same batches), the shards tile the table, the replicated parameters agree across ranks, and the trained
    table equals the single-device one."""
    from collections import defaultdict
    from utility2.utils import Data
    world, n_steps = 2, 4
    mp.spawn(_partitioned_worker, args=(world, _free_port(), str(tmp_path), str(tmp_path), n_steps), nprocs=world, join=True)
    d = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    # the single-device run of the same steps
    args, dataset, net = build_dual(weibo["data_root"])
    net = net.to(DEV)
    csr = dataset.build_adjacency()
    raw_train = weibo["raw_train"]
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2 = Data(raw_train, dataset.n_users, shuffle=False)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)
    net.train()
    rng = np.random.default_rng(5)
    for step in range(n_steps):
        users, items, labels = hub_batch(csr, N_USERS + 1, 256, rng)
        sl = _paths_for(users, by_user, 15, rng)
        opt.zero_grad()
        l1, l2 = net(t(users), t(items), t(labels), sl, train2, flag=0)
        w = net.task_weights
        (torch.exp(-2 * w[0]) * l1 + torch.exp(-2 * w[1]) * l2 + 2 * 6 * len(users) * w[0] + len(sl) * w[1]).backward()
        opt.step()
        for r in range(world):
            assert abs(d[r]["loss1"][step] - l1.item()) <= 2e-5 and abs(d[r]["loss2"][step] - l2.item()) <= 2e-5 * l2.item(), (r, step)
    assert int(d[0]["r0"]) == 0 and int(d[0]["r1"]) == int(d[1]["r0"]) and int(d[1]["r1"]) == N_USERS + 1 + N_ITEMS
    assert int(d[0]["hubs"]) + int(d[1]["hubs"]) >= 2
    for k in ("task_weights", "w"):
        assert np.abs(d[0][k] - d[1][k]).max() <= 1e-6 * max(1.0, np.abs(d[0][k]).max()), k
    table = np.concatenate([d[0]["table"], d[1]["table"]])
    want = torch.cat([net.embedding_user.weight, net.embedding_item.weight]).detach().cpu().numpy()
    assert np.abs(table - want).max() <= 0.02 * args.lr * n_steps


@pytest.mark.parametrize("fast", [True, False])
def test_one_call_partitioned_dual_step_on_the_weibo_shape_equals_the_one_gpu_step(weibo, fast):
    """spex_partitioned_dual_task_step_f32 on config 5's own shape (hub rows of > 1 024 and > 6 000 stored entries named by the batch):
    the fast path — the last forward layer at the batch's rows by spex_spmm_owned_rows_f32 (a hub row is > 100 segments), the first
    backward product pushed through the column structure (a hub COLUMN there) — and the launch-by-launch schedule against the
    single-device one-call step (DualTaskStepper, itself pinned to autograd and the fp64 oracle above): both losses of each of three
    steps, every parameter afterwards.  World size 1 (main_auto_expert_s.py:53-91, utility1/model_expert_s.py:95-168)."""
    from collections import defaultdict
    from utility2.utils import Data
    from spex_amd.dist_dual import PartitionedDualTask, PartitionedDualTaskStepper
    from spex_amd.trainer import DualTaskStepper
    args, dataset, net = build_dual(weibo["data_root"])
    _, _, core = build_dual(weibo["data_root"])                      # the same seed: the same initial parameters
    net, core = net.to(DEV), core.to(DEV)
    csr = dataset.build_adjacency()
    n_u = N_USERS + 1
    raw_train = weibo["raw_train"]
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2 = Data(raw_train, dataset.n_users, shuffle=False)
    rng = np.random.default_rng(5)
    cap = 15
    single = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr)
    model = PartitionedDualTask(core, csr, 0, 1, torch.device("cuda:0"))
    part = PartitionedDualTaskStepper(model, path_capacity=cap, path_len=train2.len_max, lr=args.lr, fast=fast)
    assert model.P.graph.n_long_rows >= 2
    for step in range(3):
        users, items, labels = hub_batch(csr, n_u, 256, rng)
        sl = _paths_for(users, by_user, cap, rng)
        inputs, mask, targets = train2.get_slice(sl)
        seq, seq_l, tgt = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(DEV) for a in (inputs, np.asarray(mask).sum(1), targets))
        single.loss_acc.zero_(); part.loss_acc.zero_()
        single.step(t(users), t(items), t(labels), seq, seq_l, tgt)
        part.step(t(users), t(items), t(labels), seq, seq_l, tgt)
        a, b = single.loss_acc.cpu().numpy(), part.loss_acc.cpu().numpy()
        assert abs(a[0] - b[0]) <= 3e-6 and abs(a[1] - b[1]) <= 2e-5 * abs(a[1]), (step, a, b)
    want = torch.cat([net.embedding_user.weight, net.embedding_item.weight]).detach()
    assert (model.E0_local.detach() - want).abs().max().item() <= 0.02 * args.lr * 3
    for (name, p_), (_, q_) in zip(net.named_parameters(), core.named_parameters()):
        if name.startswith("embedding_"):
            continue                                                 # (the partitioned model keeps its table rows in E0_local)
        assert (p_.detach() - q_.detach()).abs().max().item() <= 0.02 * args.lr * 3, name
    model.P.native.close()


def test_trust_head_split_form_survives_replay_from_a_captured_graph():
    """The split fused trust kernel replayed from a captured HIP graph (the SAME ticket tag on every replay: the path's last
    workgroup puts the ticket back to 0): three replays on changing user tables equal the eager calls bit for bit."""
    from spex_amd import _lib, ops
    from spex_amd.graph import _launch, _ptr
    n_users, T, L, H = N_USERS, 15, 6, 3
    gen = torch.Generator(device="cpu"); gen.manual_seed(8)
    tables = [(torch.rand(n_users + 1, 64, generator=gen) * 0.6 - 0.3).to(DEV) for _ in range(3)]
    params = (torch.rand(ops.trust_param_count(H, 64), generator=gen) * 0.4 - 0.2).to(DEV)
    rng = np.random.default_rng(9)
    lens = rng.integers(1, L + 1, T)
    seq = np.full((T, L), n_users, np.int64)
    for k in range(T):
        seq[k, :lens[k]] = rng.integers(0, n_users, lens[k])
    seq_d, len_d, tgt = t(seq), t(lens.astype(np.int64)), t(rng.integers(0, n_users, T))
    n_ws = int(_lib.load().spex_trust_workspace_floats(T, L, 64, H, n_users + 1))
    z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=DEV)
    table, a2, ws, ds, lb, loss, gp, gt = z(n_users + 1, 64), z(T, 64), z(n_ws), z(T, n_users), z(T), z(1), z(params.numel()), z(n_users + 1, 64)

    def call():
        _launch(DEV, "spex_trust_head_train_f32", _ptr(table), n_users + 1, _ptr(params), _ptr(seq_d), _ptr(len_d), _ptr(tgt), T, L, 64, H, 1,
                1.0, None, _ptr(a2), _ptr(ds), _ptr(lb), _ptr(ws), _ptr(loss), 0, _ptr(gp), _ptr(gt))
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        call()
    got = []
    for tb in tables:
        table.copy_(tb); gt.zero_()
        cg.replay()
        got.append((loss.clone(), lb.clone(), gp.clone(), gt.clone()))
    torch.cuda.synchronize()
    for tb, g_ in zip(tables, got):
        table.copy_(tb); gt.zero_()
        call()
        for a, b in zip((loss, lb, gp, gt), g_):
            assert torch.equal(a, b)
    assert torch.isfinite(got[0][0]).all() and got[0][0].item() > 0 and not torch.equal(got[0][2], got[1][2])


def test_trust_head_forms_agree_on_the_weibo_user_table(monkeypatch):
    """spex_trust_head_train_f32 on a 6 812-user table with 15 paths of up to 6 positions (config 5's trust batch on the Weibo
    shape): the fused kernel's one-workgroup form and its split forms (S workgroups per path) produce the same loss, path losses,
    readout vectors, parameter gradients and user-table gradient to rounding — and each repeats itself bit for bit (nothing in the
    head is atomic).  (The five-launch tiled form this test used to compare against was removed in round 4: it won at no size.)"""
    from spex_amd import _lib, ops
    from spex_amd.graph import _launch, _ptr
    n_users, T, L, H = N_USERS, 15, 6, 3
    gen = torch.Generator(device="cpu"); gen.manual_seed(3)
    table = (torch.rand(n_users + 1, 64, generator=gen) * 0.6 - 0.3).to(DEV)
    params = (torch.rand(ops.trust_param_count(H, 64), generator=gen) * 0.4 - 0.2).to(DEV)
    rng = np.random.default_rng(4)
    lens = rng.integers(1, L + 1, T)
    lens[:2] = (L, 1)
    seq = np.full((T, L), n_users, np.int64)
    for k in range(T):
        seq[k, :lens[k]] = rng.integers(0, n_users, lens[k])
    seq_d, len_d, tgt = t(seq), t(lens.astype(np.int64)), t(rng.integers(0, n_users, T))
    n_ws = int(_lib.load().spex_trust_workspace_floats(T, L, 64, H, n_users + 1))

    def run():
        z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=DEV)
        a2, ws, ds, lb, loss, gp, gt = z(T, 64), z(n_ws), z(T, n_users), z(T), z(1), z(params.numel()), z(n_users + 1, 64)
        _launch(DEV, "spex_trust_head_train_f32", _ptr(table), n_users + 1, _ptr(params), _ptr(seq_d), _ptr(len_d), _ptr(tgt), T, L, 64, H, 1,
                1.0, None, _ptr(a2), _ptr(ds), _ptr(lb), _ptr(ws), _ptr(loss), 0, _ptr(gp), _ptr(gt))
        torch.cuda.synchronize()
        return loss.clone(), lb.clone(), a2.clone(), gp.clone(), gt.clone(), ds.clone()
    fused, fused2 = run(), run()
    for a, b in zip(fused, fused2):
        assert torch.equal(a, b)
    assert torch.isfinite(fused[0]).all() and fused[0].item() > 0
    # ---- the fused kernel's SPLIT form (S workgroups per path sweep shares of the table; the last to arrive folds them in share
    #      order): every S agrees with the one-workgroup form to rounding, and REPEATS ITSELF BIT FOR BIT whichever workgroup happens
    #      to arrive last — on ONE workspace used call after call (the tickets carry a per-call tag: nothing is reset between
    #      calls), whose first contents are random bits (the library may not assume a zeroed workspace).
    junk = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_ws,), dtype=torch.int32, device=DEV).view(torch.float32)
    z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=DEV)

    def run_on(ws, split):
        monkeypatch.setenv("SPEX_TRUST_SPLIT", str(split))
        a2, ds, lb, loss, gp, gt = z(T, 64), z(T, n_users), z(T), z(1), z(params.numel()), z(n_users + 1, 64)
        _launch(DEV, "spex_trust_head_train_f32", _ptr(table), n_users + 1, _ptr(params), _ptr(seq_d), _ptr(len_d), _ptr(tgt), T, L, 64, H, 1,
                1.0, None, _ptr(a2), _ptr(ds), _ptr(lb), _ptr(ws), _ptr(loss), 0, _ptr(gp), _ptr(gt))
        return loss, lb, a2, gp, gt, ds
    one = run_on(junk.clone(), 1)
    for nm, a, b in zip(("loss", "path losses", "a2", "grad params", "grad table", "d scores"), one, fused):     # (`fused`: the library's own S)
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) <= 2e-5, nm
    for split in (2, 4, 8):
        ws = junk.clone()
        runs = [run_on(ws, split) for _ in range(12)]
        torch.cuda.synchronize()
        for r in runs[1:]:
            for nm, a, b in zip(("loss", "path losses", "a2", "grad params", "grad table", "d scores"), runs[0], r):
                assert torch.equal(a, b), (split, nm)
        assert torch.equal(runs[0][2], one[2])                           # a2: the forward chain does not depend on the split
        for nm, a, b in zip(("loss", "path losses", "a2", "grad params", "grad table", "d scores"), one, runs[0]):
            assert rel_err(b.cpu().numpy(), a.cpu().numpy()) <= 2e-5, (split, nm)
