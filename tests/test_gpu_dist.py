"""Row-partitioned propagation on the GPU with two ranks sharing the one card of the test box (gloo moves the
all-gathered rows; on a real node the same code runs one rank per GPU over RCCL).  Each rank's HIP SpMM on its row
block + the padded all-gather must reproduce the single-device result bit for bit, forward and backward."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu


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
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.dist import PartitionedLightGCN
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    tr = load_epinion2()["train"]
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    uw, iw = epinion2_tables(3186, 12407)
    E0 = np.concatenate([uw, iw])
    P = PartitionedLightGCN(*csr, 3186, 3, 64, rank, world,
                            lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev)
    lo = P.propagate(torch.from_numpy(E0[P.r0:P.r1].copy()).to(dev)).clone()
    g = torch.from_numpy(E0[::-1].copy()[P.r0:P.r1].copy()).to(dev)
    grad = P.propagate_bwd(g).clone()
    full = P.gather_output()
    u = torch.arange(0, 3185, 7, device=dev)
    i = torch.arange(0, 12407, 31, device=dev)[: len(u)]
    pu, pi = P.padded_index(u, i)
    # the one-launch exchange helpers agree with the plan-based tensor ops
    pos = torch.cat([pu, pi, pu[:5]])                                   # (positions repeat)
    a = P.fetch_rows(P.plan_rows(pos), torch.empty(len(pos), 64, device=dev)).clone()
    b = P.fetch_rows_at(pos, torch.empty(len(pos), 64, device=dev))
    assert torch.equal(a, b)
    upd = torch.randn(len(pos), 64, device=dev)
    t1, t2 = torch.zeros(P.n_local, 64, device=dev), torch.zeros(P.n_local, 64, device=dev)
    own_idx, local = P.plan_rows(pos)
    t1.index_add_(0, local, upd.index_select(0, own_idx))
    P.add_owned_rows(upd, pos, t2, clear=True)
    assert (t1 - t2).abs().max().item() <= 1e-5 and torch.count_nonzero(upd).item() == 0
    # three exact training steps (BCE + backward + Adam) on the partitioned model
    from spex_amd.dist import PartitionedStepper
    E0_local = torch.from_numpy(E0[P.r0:P.r1].copy()).to(dev)
    st = PartitionedStepper(P, E0_local, lr=1e-3)
    rng = np.random.default_rng(3)
    losses = []
    for _ in range(3):
        bu = torch.from_numpy(rng.integers(0, 3185, 256)); bi = torch.from_numpy(rng.integers(0, 12407, 256))
        by = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32))
        losses.append(st.step_bce(bu, bi, by).item())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo.cpu().numpy(), grad=grad.cpu().numpy(), r0=P.r0, r1=P.r1,
             fu=full[pu].cpu().numpy(), fi=full[pi].cpu().numpy(), u=u.cpu().numpy(), i=i.cpu().numpy(),
             trained=E0_local.cpu().numpy(), losses=np.asarray(losses))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_gpu_match_single_device(tmp_path):
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    tr = load_epinion2()["train"]
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    uw, iw = epinion2_tables(3186, 12407)
    E0 = np.concatenate([uw, iw])
    g = SpexGraph(*csr)
    ref = g.propagate(torch.from_numpy(E0).cuda(), 3).cpu().numpy()
    ref_grad = g.propagate_bwd(torch.from_numpy(E0[::-1].copy()).cuda(), 3).cpu().numpy()
    lo, grad = np.zeros_like(ref), np.zeros_like(ref)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        lo[int(d["r0"]):int(d["r1"])] = d["lo"]
        grad[int(d["r0"]):int(d["r1"])] = d["grad"]
        assert np.array_equal(d["fu"], ref[d["u"]]) and np.array_equal(d["fi"], ref[3186 + d["i"]])
    assert np.array_equal(lo, ref)          # a row's summation order does not depend on the partition
    assert np.array_equal(grad, ref_grad)
    # the same three training steps on one device
    from spex_amd.trainer import LightGCNStepper
    st = LightGCNStepper(g, torch.from_numpy(E0.copy()).cuda(), 3186, n_layers=3, lr=1e-3)
    rng = np.random.default_rng(3)
    ref_losses = []
    for _ in range(3):
        bu = torch.from_numpy(rng.integers(0, 3185, 256)).cuda(); bi = torch.from_numpy(rng.integers(0, 12407, 256)).cuda()
        by = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32)).cuda()
        ref_losses.append(st.step_bce(bu, bi, by).item())
    trained = np.zeros_like(ref)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        trained[int(d["r0"]):int(d["r1"])] = d["trained"]
        assert np.allclose(d["losses"], ref_losses, rtol=0, atol=2e-6)
    want = st.E0.cpu().numpy()
    assert np.abs(trained - want).max() <= 5e-6 * np.abs(want).max()


def _nccl_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.dist import PartitionedLightGCN, PartitionedStepper
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    tr = load_epinion2()["train"]
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    uw, iw = epinion2_tables(3186, 12407)
    E0 = torch.from_numpy(np.concatenate([uw, iw])).to(dev)
    P = PartitionedLightGCN(*csr, 3186, 3, 64, rank, world,
                            lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev,
                            always_collective=True)
    lo = P.propagate(E0.clone()).clone()
    st = PartitionedStepper(P, E0.clone(), lr=1e-3)
    rng = np.random.default_rng(3)
    bu = torch.from_numpy(rng.integers(0, 3185, 256)); bi = torch.from_numpy(rng.integers(0, 12407, 256))
    by = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32))
    loss = st.step_bce(bu, bi, by).item()
    dist.barrier()
    np.savez(os.path.join(out_dir, "nccl.npz"), lo=lo.cpu().numpy(), loss=loss)
    dist.destroy_process_group()


def test_rccl_backend_smoke_world_size_one(tmp_path):
    """The collectives of the partitioned schedule issued through the real backend (`nccl` == RCCL) on the one GPU of
    the test box: a one-rank all-gather / all-reduce / barrier on device tensors.  (Multi-rank RCCL needs one GPU per
    rank; the N-rank schedule itself is covered with gloo above and in tests/test_dist_cpu.py.)"""
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    d = np.load(tmp_path / "nccl.npz")
    tr = load_epinion2()["train"]
    g = SpexGraph(*lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407))
    uw, iw = epinion2_tables(3186, 12407)
    ref = g.propagate(torch.from_numpy(np.concatenate([uw, iw])).cuda(), 3).cpu().numpy()
    assert np.array_equal(d["lo"], ref)
    assert np.isfinite(float(d["loss"]))
