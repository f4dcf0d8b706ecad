"""The multi-GPU schedule (1-D row partition + all-gather per layer) exercised without GPUs: world_size-2 gloo
processes on the CPU, with the local SpMM replaced by an oracle-backed stand-in that has SpexGraph's `.spmm`
signature.  What is under test is spex_amd/dist.py: partition bounds, the padded column remap, the all-gather
plumbing, the forward/backward layer schedule and the owner-computes index translation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO


class OracleGraph:
    """CPU stand-in for SpexGraph (test double): same .spmm contract, computed by the oracle."""

    def __init__(self, rowptr, col, val, n_cols=None):
        from oracle import oracle as O
        self.O, self.csr = O, (rowptr, col, val)
        self.n_rows, self.n_cols, self.nnz = len(rowptr) - 1, n_cols, len(col)

    def spmm(self, X, Y=None, add_in=None, add_div=1.0, acc_in=None, acc_out=None, acc_div=1.0):
        assert X.shape[0] == self.n_cols
        y = self.O.spmm(*self.csr, X.numpy())
        if add_in is not None:
            y = y + add_in.numpy() / np.float32(add_div)
        if Y is not None:
            Y.copy_(torch.from_numpy(y))
        if acc_out is not None:
            acc_out.copy_(torch.from_numpy((acc_in.numpy() + y) / np.float32(acc_div)))
        return Y if Y is not None else acc_out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spex_amd.dist import PartitionedLightGCN
    g = np.load(os.path.join(GOLDEN, "lightgcn_tiny.npz"))
    csr = (g["rowptr"], g["col"], g["val"])
    n_u = int(g["n_user"]) + 1
    P = PartitionedLightGCN(*csr, n_u, 3, 64, rank, world, OracleGraph, "cpu")
    E0 = torch.from_numpy(g["E0"][P.r0:P.r1].copy())
    lo = P.propagate(E0).clone()
    full = P.gather_output().clone()
    gl = torch.from_numpy(g["E0"][::-1].copy()[P.r0:P.r1].copy())     # any deterministic upstream gradient
    grad = P.propagate_bwd(gl).clone()
    u = torch.from_numpy(g["batch_users"][0]); i = torch.from_numpy(g["batch_items"][0])
    pu, pi = P.padded_index(u, i)
    P.propagate(E0)                                   # fetch_rows reads the rank's current light_out
    pos = torch.cat([pu, pi])
    fetched = P.fetch_rows(P.plan_rows(pos), torch.empty(len(pos), 64))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo.numpy(), grad=grad.numpy(), r0=P.r0, r1=P.r1,
             full_u=full[pu].numpy(), full_i=full[pi].numpy(), own=P.own_slice(full).numpy(), fetched=fetched.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_partitioned_schedule_matches_single_device(tmp_path, oracle, golden, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = golden("lightgcn_tiny")
    csr = (g["rowptr"], g["col"], g["val"])
    n_u = int(g["n_user"]) + 1
    ref = oracle.propagate_mean(*csr, g["E0"], 3)
    gl = g["E0"][::-1].copy() / np.float32(4.0)
    G = gl.copy()
    t = oracle.csr_transpose(*csr, len(csr[0]) - 1)
    for _ in range(3):
        G = gl + oracle.spmm(t[0], t[1], t[2], G)
    got_lo, got_grad = np.zeros_like(ref), np.zeros_like(ref)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        got_lo[int(d["r0"]):int(d["r1"])] = d["lo"]
        got_grad[int(d["r0"]):int(d["r1"])] = d["grad"]
        # every rank sees the same gathered table, addressed through the padded layout
        assert np.array_equal(d["full_u"], ref[g["batch_users"][0]])
        assert np.array_equal(d["full_i"], ref[n_u + g["batch_items"][0]])
        assert np.array_equal(d["own"], ref[int(d["r0"]):int(d["r1"])])
        assert np.array_equal(d["fetched"], np.concatenate([ref[g["batch_users"][0]], ref[n_u + g["batch_items"][0]]]))
    # partitioning does not change any row's summation order: bit-identical to the single-device result
    assert np.array_equal(got_lo, ref) and np.array_equal(got_lo, g["light_out"])
    assert np.array_equal(got_grad, G)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_row_partition_layout(golden, epinion2, world):
    from spex_amd.dist import RowPartition, balanced_row_bounds
    from spex_amd.graph import lightgcn_norm_adj
    tr = epinion2["train"]
    rowptr, col, val = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    p = RowPartition(rowptr, world)
    assert p.bounds[0] == 0 and p.bounds[-1] == len(rowptr) - 1 and (np.diff(p.bounds) >= 0).all()
    nnz_share = np.diff(rowptr[p.bounds])
    assert nnz_share.sum() == len(col)
    if world > 1:
        rows = np.diff(p.bounds)                                          # default bounds: the exchange dominates -> rows equalised
        assert rows.max() <= 1.12 * rows.mean() + 8                       # (the padded all-gather carries world * max_rows rows)
        nnz4 = np.diff(rowptr[balanced_row_bounds(rowptr, world, row_cost=4)])
        assert nnz4.max() <= 1.25 * nnz4.mean() + 1100                    # row_cost=4: balanced by stored entries (max row 1020)
    g = np.arange(len(rowptr) - 1)
    pos = p.to_padded(g)
    assert len(np.unique(pos)) == len(g) and pos.max() < p.n_padded
    assert np.array_equal(p.to_padded_torch(torch.from_numpy(g)).numpy(), pos)
    # local blocks tile the matrix and their remapped columns address the same rows
    total = 0
    for r in range(world):
        lr, lc, lv, _ = p.local_block(rowptr, col, val, r)
        b, e = rowptr[p.bounds[r]], rowptr[p.bounds[r + 1]]
        assert np.array_equal(lc, p.to_padded(col[b:e])) and np.array_equal(lv, val[b:e])
        assert (np.diff(lc.astype(np.int64).reshape(-1))[np.diff(np.repeat(np.arange(len(lr) - 1), np.diff(lr))) == 0] > 0).all()
        total += len(lc)
    assert total == len(col)


def test_serial_emulation_of_all_ranks_is_bit_identical(oracle, golden):
    """P in {1,2,4,8} emulated serially (world of one process acting as each rank in turn on a shared gathered
    buffer): per-row results do not depend on P."""
    from spex_amd.dist import RowPartition
    g = golden("lightgcn_tiny")
    csr = (g["rowptr"], g["col"], g["val"])
    ref = g["light_out"]
    for world in (1, 2, 4, 8):
        p = RowPartition(csr[0], world)
        blocks = [OracleGraph(*p.local_block(*csr, r)[:3], n_cols=p.n_padded) for r in range(world)]
        cur = g["E0"].copy()
        acc = cur.copy()
        for l in range(3):
            gathered = np.zeros((p.n_padded, 64), np.float32)
            gathered[p.to_padded(np.arange(len(cur)))] = cur
            nxt = np.zeros_like(cur)
            for r in range(world):
                r0, r1 = p.bounds[r], p.bounds[r + 1]
                Y = torch.zeros(r1 - r0, 64)
                blocks[r].spmm(torch.from_numpy(gathered), Y=Y)
                nxt[r0:r1] = Y.numpy()
            acc = acc + nxt
            cur = nxt
        assert np.array_equal(acc / np.float32(4.0), ref)


# ---------------------------------------------------------------------------------------------- BASELINE config 3 shape
class OracleOps:
    """CPU stand-in for spex_amd.ops in the schedule tests (test double): score_bce and adam_step with the kernels'
    contracts, computed by the oracle."""

    def __init__(self):
        from oracle import oracle as O
        self.O = O

    def score_bce(self, users_tab, items_tab, u_idx, i_idx, labels=None, grad_users=None, grad_items=None, grad_scale=0.0,
                  loss_sum=None, want_gamma=True):
        gamma, loss, gu, gi = self.O.score_bce(users_tab.numpy(), items_tab.numpy(), u_idx.numpy(), i_idx.numpy(),
                                               labels.numpy().astype(np.float32), want_grad=True)
        B = len(gamma)
        # the oracle returns d(mean loss)/d rows; the kernel contract is grad += (sigmoid - y) * grad_scale * other row
        grad_users += torch.from_numpy(gu) * (grad_scale * B)
        grad_items += torch.from_numpy(gi) * (grad_scale * B)          # (may be the same table: both parts accumulate)
        s = torch.tensor([float(loss) * B], dtype=torch.float32)
        if loss_sum is not None:
            loss_sum += s
            return None, loss_sum
        return torch.from_numpy(gamma), s

    def adam_step(self, p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, zero=None):
        self.O.adam_step(p.numpy(), g.numpy(), m.numpy(), v.numpy(), t, lr, beta1, beta2, eps)
        if zero is not None:
            zero.zero_()


def weibo_like(n_users=6812, n_items=20000, n_edges=400000):
    """The config-3/5 stand-in graph (Weibo's published user count, Trust_SPEX/code/main_trust.py:42; raw data are not in
    the reference): heavy-tailed, hubs beyond 1 024 entries, empty rows."""
    from spex_amd.datasets import synthetic_interactions, xavier_uniform_np
    from spex_amd.graph import lightgcn_norm_adj
    u, i = synthetic_interactions(n_users, n_items, n_edges, seed=7)
    csr = lightgcn_norm_adj(u.numpy(), i.numpy(), n_users, n_items)
    E0 = xavier_uniform_np(len(csr[0]) - 1, 64, np.random.default_rng(1))
    return csr, E0, n_users + 1


def _weibo_batches(n_users, n_items, steps, B=256):
    rng = np.random.default_rng(3)
    return [(rng.integers(0, n_users, B), rng.integers(0, n_items, B), (rng.random(B) < 1 / 6).astype(np.float32))
            for _ in range(steps)]


def _weibo_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from spex_amd.dist import PartitionedLightGCN, PartitionedStepper, balanced_row_bounds
    csr, E0, n_u = weibo_like()
    # bounds balanced by stored entries (row_cost=4): user shards short, item shards long — the padded layout under skew
    P = PartitionedLightGCN(*csr, n_u, 3, 64, rank, world, OracleGraph, "cpu", bounds=balanced_row_bounds(csr[0], world, row_cost=4))
    E0_local = torch.from_numpy(E0[P.r0:P.r1].copy())
    lo = P.propagate(E0_local).clone()
    grad = P.propagate_bwd(torch.from_numpy(E0[::-1].copy()[P.r0:P.r1].copy())).clone()
    st = PartitionedStepper(P, E0_local, lr=1e-3, ops=OracleOps())
    losses = []
    for bu, bi, by in _weibo_batches(6812, 20000, 2):
        losses.append(float(st.step_bce(torch.from_numpy(bu), torch.from_numpy(bi), torch.from_numpy(by))))
    assert torch.count_nonzero(st.g_local).item() == 0 and torch.count_nonzero(st.grad_rows).item() == 0
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo.numpy(), grad=grad.numpy(), r0=P.r0, r1=P.r1,
             trained=E0_local.numpy(), losses=np.asarray(losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_weibo_shaped_graph_partitioned_forward_backward_and_training_step(tmp_path, oracle, world):
    """BASELINE config 3's shape through the real collectives (gloo, `world` processes): rows of > 1 024 entries cut by
    shard boundaries, empty rows, skewed shards.  Forward and backward are bit-identical to one device; two exact
    training steps (BCE + backward + Adam through PartitionedStepper) agree with the single-device restatement to the
    re-association of the batch's duplicate-row gradient sums."""
    mp.spawn(_weibo_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    csr, E0, n_u = weibo_like()
    deg = np.diff(csr[0])
    assert deg.max() > 1024 and (deg == 0).sum() > 0
    ref = oracle.propagate_mean(*csr, E0, 3, n_threads=4)
    gl = E0[::-1].copy() / np.float32(4.0)
    G = gl.copy()
    tcsr = oracle.csr_transpose(*csr, len(deg))
    for _ in range(3):
        G = gl + oracle.spmm(*tcsr, G, n_threads=4)
    # the same two training steps on one device
    W, m, v = E0.copy(), np.zeros_like(E0), np.zeros_like(E0)
    want_losses = []
    for t_, (bu, bi, by) in enumerate(_weibo_batches(6812, 20000, 2), 1):
        _, loss, grad = oracle.lightgcn_loss_and_grad(*csr, W, n_u, 3, bu, bi, by, n_threads=4)
        oracle.adam_step(W, grad, m, v, t_)
        want_losses.append(float(loss))
    lo, grad_got, trained = np.zeros_like(ref), np.zeros_like(ref), np.zeros_like(ref)
    shard_rows = []
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        r0, r1 = int(d["r0"]), int(d["r1"])
        lo[r0:r1], grad_got[r0:r1], trained[r0:r1] = d["lo"], d["grad"], d["trained"]
        shard_rows.append(r1 - r0)
        assert np.allclose(d["losses"], want_losses, rtol=0, atol=2e-6)
    assert max(shard_rows) > 1.5 * min(shard_rows)           # shards balanced by entries are skewed in rows
    assert np.array_equal(lo, ref) and np.array_equal(grad_got, G)
    assert np.abs(trained - W).max() <= 5e-6 * np.abs(W).max()


def test_weibo_shaped_graph_serial_emulation_of_8_ranks(oracle):
    """P = 8 on the same graph without 8 processes: every rank's row block + padded column remap evaluated in turn on
    one shared gathered buffer (forward and backward layer), bit-identical to the unpartitioned product; hub rows end up
    on several different shards and at least one shard boundary cuts the user block."""
    from spex_amd.dist import RowPartition
    csr, E0, n_u = weibo_like()
    tcsr = oracle.csr_transpose(*csr, len(csr[0]) - 1)
    for mat, X in ((csr, E0), (tcsr, E0[::-1].copy())):
        want = oracle.spmm(*mat, X, n_threads=4)
        p = RowPartition(mat[0], 8)
        assert 0 < np.searchsorted(p.bounds, n_u) < 8
        gathered = np.zeros((p.n_padded, 64), np.float32)
        gathered[p.to_padded(np.arange(len(X)))] = X
        got = np.zeros_like(want)
        hubs_on = set()
        for r in range(8):
            lr, lc, lv, _ = p.local_block(*mat, r)
            got[p.bounds[r]:p.bounds[r + 1]] = oracle.spmm(lr, lc, lv, gathered, n_threads=4)
            if (np.diff(lr) > 1024).any():
                hubs_on.add(r)
        assert np.array_equal(got, want) and len(hubs_on) >= 2


# ---------------------------------------------------------------------------------------------- BASELINE config 5: partitioned dual-task step
class TorchKernels:
    """CPU stand-ins (test doubles, torch ops with autograd) for the kernels the partitioned dual-task model calls through
    its `kernels` namespace: the two-expert gate (model_expert_s.py:156-161) and gather.dot.BCE (model.py:115-120)."""

    @staticmethod
    def expert_gate_autograd(raw, prop, att):
        a = torch.softmax(torch.cat([raw, prop], 1) @ att, 1)
        return raw * a[:, :1] + prop * a[:, 1:2]

    class ScoreBCELoss:
        @staticmethod
        def apply(table, n_user_rows, u_idx, i_idx, labels):
            gamma = (table[:n_user_rows][u_idx] * table[n_user_rows:][i_idx]).sum(1)
            return torch.nn.functional.binary_cross_entropy_with_logits(gamma, labels)


class _HostOnlyDataset:
    """The Loader's data without its device graph (the partitioned model builds per-rank row blocks itself)."""

    def __init__(self, n_users, m_items):
        self.n_users, self.m_items = n_users, m_items

    def getSparseGraph(self):
        return None


def _install_cpu_path_attention():
    """Route the trust head's path attention (HIP kernels in the product) to the oracle's torch restatement."""
    from oracle import trust_oracle
    from spex_amd import ops
    import utility2.layers as layers
    ops.trust_head_supported = lambda d, L, n_heads: False          # the fused trust head is HIP-only: layer-by-layer form here
    ops.path_attention = lambda src, seq, seq_l, a, positional: torch.cat(
        [trust_oracle.path_attention(src, seq, seq_l, a[h], True) for h in range(a.shape[0])], dim=2)
    layers.GraphAttentionLayer.forward = lambda self, emb, seq, seq_l: trust_oracle.path_attention(emb, seq, seq_l, self.a, self.concat)


def _dual_core(g11):
    import lg_parser
    import utility1.model_expert_s as mex
    import utility1.utils as utils
    args = lg_parser.parse_args_r(["--dataset", "tiny"])
    utils.set_seed(args.seed)
    core = mex.LightGCN(args, _HostOnlyDataset(int(g11["n_users"]), 60))
    sd = {k[6:].replace("__", "."): torch.from_numpy(g11[k]) for k in g11.files if k.startswith("state_")}
    core.load_state_dict(sd)                                  # the reference's parameters (G11)
    core._fuse_tables()
    return args, core


def _dual_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "spex_amd", "dropin"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    _install_cpu_path_attention()
    from spex_amd.dist_dual import PartitionedDualTask
    from utility2.utils import Data
    g11, lg = np.load(os.path.join(GOLDEN, "trust_tiny.npz")), np.load(os.path.join(GOLDEN, "lightgcn_tiny.npz"))
    args, core = _dual_core(g11)
    csr = (lg["rowptr"], lg["col"], lg["val"])
    model = PartitionedDualTask(core, csr, rank, world, "cpu", graph_factory=OracleGraph, kernels=TorchKernels)
    lens = g11["train_mask"].sum(1)
    train = Data(([r[:l].tolist() for r, l in zip(g11["train_inputs"], lens)], g11["train_targets"].tolist()),
                 int(g11["n_users"]), shuffle=False)
    bu, bi, bl = (torch.from_numpy(lg[k][0]) for k in ("batch_users", "batch_items", "batch_labels"))
    loss1, loss2 = model(bu, bi, bl, g11["slice_indices"], train)
    (loss1 + loss2).backward()
    model.reduce_gate_gradients()
    grads = {n.replace(".", "__"): p.grad.numpy() for n, p in core.named_parameters() if p.grad is not None}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss1=loss1.item(), loss2=loss2.item(), r0=model.P.r0, r1=model.P.r1,
             g_table=model.E0_local.grad.numpy(), **{"grad_" + k: v for k, v in grads.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_partitioned_dual_task_step_matches_the_reference_golden(tmp_path, golden, world):
    """BASELINE config 5's multi-GPU form (spex_amd.dist_dual) through real collectives (gloo): gate on local rows,
    replicated rec batch exchanged owner-computes, trust head on the all-gathered E0 user block, gate gradients
    all-reduced.  With the reference's parameters (G11) both losses and EVERY gradient equal the golden minted from the
    reference's single-process model — for 1, 2 and 3 ranks (3: a shard boundary inside the user block and one inside the
    item block)."""
    mp.spawn(_dual_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = golden("trust_tiny")
    n_u = int(g["n_users"]) + 1
    want_table = np.concatenate([g["grad_embedding_user__weight"], g["grad_embedding_item__weight"]])
    got_table = np.zeros_like(want_table)
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert abs(float(d["loss1"]) - float(g["loss1"])) <= 2e-6 and abs(float(d["loss2"]) - float(g["loss2"])) <= 2e-5
        got_table[int(d["r0"]):int(d["r1"])] = d["g_table"]
        checked = 0
        for k in g.files:
            if k.startswith("grad_") and "embedding" not in k:
                assert rel(d[k], g[k]) <= 5e-5, (r, k)          # replicated parameters: the full gradient on every rank
                checked += 1
        assert checked >= 12
    assert rel(got_table, want_table) <= 5e-5
