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
        assert nnz_share.max() <= 1.25 * nnz_share.mean() + 1100          # balanced by stored entries (max row 1020)
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
