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
    # ... and against the CPU oracle, not only against the one-rank HIP result (the single-device gate of test_gpu_kernels.py:
    # rows cut into segments re-associate <= 16 partial sums)
    from oracle import oracle as O
    want_lo = O.propagate_mean(*csr, E0, 3, n_threads=4)
    rel_err = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
    assert rel_err(lo, want_lo) <= 1e-6
    tcsr = O.csr_transpose(*csr, len(E0))
    gl = E0[::-1].copy() / np.float32(4.0)
    G_ = gl.copy()
    for _ in range(3):
        G_ = gl + O.spmm(*tcsr[:3], G_, n_threads=4)
    assert rel_err(grad, G_) <= 2e-6
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


# ---------------------------------------------------------------------------------------------- BASELINE config 3 shape
def _weibo():
    from spex_amd.datasets import synthetic_interactions, xavier_uniform_np
    from spex_amd.graph import lightgcn_norm_adj
    u, i = synthetic_interactions(6812, 20000, 400000, seed=7)
    csr = lightgcn_norm_adj(u.numpy(), i.numpy(), 6812, 20000)
    return csr, xavier_uniform_np(len(csr[0]) - 1, 64, np.random.default_rng(1)), 6813


def _weibo_batches(steps, B=256):
    rng = np.random.default_rng(3)
    return [(rng.integers(0, 6812, B), rng.integers(0, 20000, B), (rng.random(B) < 1 / 6).astype(np.float32)) for _ in range(steps)]


def _weibo_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spex_amd.dist import PartitionedLightGCN, PartitionedStepper, balanced_row_bounds
    from spex_amd.graph import SpexGraph
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    csr, E0, n_u = _weibo()
    # bounds balanced by stored entries (row_cost=4): shards skewed in rows — the padded layout under skew
    P = PartitionedLightGCN(*csr, n_u, 3, 64, rank, world,
                            lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev,
                            bounds=balanced_row_bounds(csr[0], world, row_cost=4))
    E0_local = torch.from_numpy(E0[P.r0:P.r1].copy()).to(dev)
    lo = P.propagate(E0_local).clone()
    grad = P.propagate_bwd(torch.from_numpy(E0[::-1].copy()[P.r0:P.r1].copy()).to(dev)).clone()
    st = PartitionedStepper(P, E0_local, lr=1e-3)
    acc = torch.zeros(1, device=dev)
    batches = _weibo_batches(4)
    losses = [st.step_bce(torch.from_numpy(bu).to(dev), torch.from_numpy(bi).to(dev), torch.from_numpy(by).to(dev)).item()
              for bu, bi, by in batches[:2]]
    pos = [st.positions(torch.from_numpy(bu), torch.from_numpy(bi)) for bu, bi, _ in batches[2:]]       # precomputed form
    before = torch.cuda.memory_allocated()
    for (bu, bi, by), p in zip(batches[2:], pos):
        st.step_bce(torch.from_numpy(bu).to(dev), torch.from_numpy(bi).to(dev), torch.from_numpy(by).to(dev), pos=p, loss_acc=acc)
    assert torch.cuda.memory_allocated() <= before + 3 * 256 * 8 + 4096          # nothing but the uploaded batch
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo.cpu().numpy(), grad=grad.cpu().numpy(), r0=P.r0, r1=P.r1,
             trained=E0_local.cpu().numpy(), losses=np.asarray(losses), acc=acc.item(), hubs=P.graph.n_long_rows)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_gpu_weibo_shaped_graph(tmp_path):
    """BASELINE config 3's shape (6 812 users, hubs beyond 1 024 entries cut across shards, empty rows) row-partitioned
    over two ranks on the GPU: forward and backward bit-identical to the single-device HIP result, four exact training
    steps through the launch-only PartitionedStepper (positions precomputed for the last two) within 5e-6 of the
    single-device stepper."""
    from spex_amd.graph import SpexGraph
    from spex_amd.trainer import LightGCNStepper
    world = 2
    mp.spawn(_weibo_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    csr, E0, n_u = _weibo()
    g = SpexGraph(*csr)
    ref = g.propagate(torch.from_numpy(E0).cuda(), 3).cpu().numpy()
    ref_grad = g.propagate_bwd(torch.from_numpy(E0[::-1].copy()).cuda(), 3).cpu().numpy()
    st = LightGCNStepper(g, torch.from_numpy(E0.copy()).cuda(), n_u, n_layers=3, lr=1e-3)
    ref_losses = [st.step_bce(torch.from_numpy(bu).cuda(), torch.from_numpy(bi).cuda(), torch.from_numpy(by).cuda()).item()
                  for bu, bi, by in _weibo_batches(4)]
    lo, grad, trained = np.zeros_like(ref), np.zeros_like(ref), np.zeros_like(ref)
    hubs = 0
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        r0, r1 = int(d["r0"]), int(d["r1"])
        lo[r0:r1], grad[r0:r1], trained[r0:r1] = d["lo"], d["grad"], d["trained"]
        assert np.allclose(d["losses"], ref_losses[:2], rtol=0, atol=2e-6)
        assert abs(float(d["acc"]) - 256 * (ref_losses[2] + ref_losses[3])) <= 2e-3
        hubs += int(d["hubs"] > 0)
    assert hubs == world                                  # long rows on both shards
    assert np.array_equal(lo, ref) and np.array_equal(grad, ref_grad)
    want = st.E0.cpu().numpy()
    assert np.abs(trained - want).max() <= 5e-6 * np.abs(want).max()


# ---------------------------------------------------------------------------------------------- BASELINE config 5: partitioned dual-task
def _dual_worker(rank, world, port, out_dir, data_root, n_steps):
    import random
    import sys
    from collections import defaultdict
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "spex_amd", "dropin"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.utils.data import DataLoader
    import lg_parser
    import utility1.dataloader as dl
    import utility1.model_expert_s as mex
    import utility1.utils as utils
    from utility2.utils import Data
    from spex_amd.dist_dual import PartitionedDualTask
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    t = np.load(os.path.join(GOLDEN, "trust_epinion2_paths.npz"))
    raw_train = ([r[:l].tolist() for r, l in zip(t["train_paths"].astype(np.int64), t["train_len"])],
                 t["train_targets"].astype(np.int64).tolist())
    args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", data_root])
    utils.set_seed(args.seed)                                             # every rank: the same seeds => the same batches
    dataset = dl.Loader(args)
    loader = DataLoader(dl.LightTrainData(dataset.rec_train_data, dataset.m_item, dataset.train_mat), batch_size=256, shuffle=True)
    by_user = defaultdict(list)
    for k, p in enumerate(raw_train[0]):
        by_user[p[0]].append(k)
    train2 = Data(raw_train, dataset.n_users, shuffle=False)
    cap = 3 * (len(raw_train[0]) // len(loader))
    core = mex.LightGCN(args, dataset).to(dev)
    model = PartitionedDualTask(core, dataset.build_adjacency(), rank, world, dev)
    opt = torch.optim.Adam(model.trained_parameters(), lr=args.lr)
    loader.dataset.ng_sample()
    core.train()
    l1s, l2s, n_paths = [], [], []
    for step, (user, item, label) in enumerate(loader):
        if step == n_steps:
            break
        opt.zero_grad()
        chosen = []
        for u in set(user.numpy().tolist()):
            chosen.extend(by_user[u])
        if len(chosen) > cap:
            chosen = random.sample(chosen, cap)
        l1, l2 = model(user, item, label, np.array(chosen, dtype=int), train2)
        w = model.task_weights
        (torch.exp(-2 * w[0]) * l1 + torch.exp(-2 * w[1]) * l2 + 2 * 6 * len(user) * w[0] + len(chosen) * w[1]).backward()
        model.reduce_gate_gradients()
        opt.step()
        l1s.append(l1.item()); l2s.append(l2.item()); n_paths.append(len(chosen))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss1=np.asarray(l1s), loss2=np.asarray(l2s), n_paths=np.asarray(n_paths),
             r0=model.P.r0, r1=model.P.r1, table=model.E0_local.detach().cpu().numpy(),
             task_weights=model.task_weights.detach().cpu().numpy(), att_exp1=core.att_exp1.detach().cpu().numpy(),
             w=core.w.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_dual_task_training_matches_the_reference_epinion2(tmp_path, golden):
    """BASELINE config 5 (dual-task, row-partitioned) with two ranks on the GPU, on Epinion2 + the reference-minted trust
    paths: the first 16 training steps of main_auto_expert_s.py (same seeds on both ranks => same batches and paths)
    reproduce the REFERENCE's per-step losses of both tasks (golden G13, minted from the reference's single-process run),
    and both ranks hold identical replicated parameters afterwards."""
    from spex_amd.datasets import materialise_epinion2
    g = golden("dual_epinion2_epochs")
    root = materialise_epinion2(str(tmp_path / "data"))
    world, n_steps = 2, len(g["loss1_first"])
    mp.spawn(_dual_worker, args=(world, _free_port(), str(tmp_path), root, n_steps), nprocs=world, join=True)
    d = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in range(world):
        assert np.array_equal(d[r]["n_paths"], g["n_paths"][:n_steps].astype(int))
        assert np.abs(d[r]["loss1"] - g["loss1_first"]).max() <= 2e-5, (r, d[r]["loss1"], g["loss1_first"])
        assert (np.abs(d[r]["loss2"] - g["loss2_first"]) <= 1e-4 * g["loss2_first"]).all(), r
    for k in ("task_weights", "att_exp1", "w"):
        assert np.abs(d[0][k] - d[1][k]).max() <= 1e-6 * max(1.0, np.abs(d[0][k]).max()), k
    assert int(d[0]["r1"]) == int(d[1]["r0"]) and int(d[1]["r1"]) == 3186 + 12407


def test_bench_multi_rank_path_rehearsal():
    """bench.py's N > 1 code path exactly as the driver launches it (python -m torch.distributed.run, one process per
    rank), rehearsed with two ranks sharing the test box's one GPU (SPEX_BENCH_SHARE_GPU=1, collectives through gloo;
    the driver's runs use one GPU per rank over RCCL): it must print one well-formed JSON line with the whole-job
    figures."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, SPEX_BENCH_SHARE_GPU="1", SPEX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--hbm-log2-nodes", "17"]                    # (the strong-scaling leg on a 2^17-node graph here; 2^24 in the driver's runs)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["scaling"] == "weak" and out["unit"] == "edges/s"
    assert out["config"]["parallelism"] == "row-partition x2" and out["value"] > 0 and out["roofline"]["achieved"] > 0
    # whole-job aggregate: L * nnz(Epinion2 x 2) * steps / time
    assert abs(out["value"] - 3 * 2 * 418608 * 20 / (out["ms_per_step"] * 20 * 1e-3)) <= 1e-6 * out["value"]
    # the strong-scaling leg: ONE graph (here 2^17 nodes) row-partitioned over the two ranks
    ss = out["extra"]["strong_scaling_hbm_graph"]
    assert "error" not in ss, ss
    assert ss["propagation_ms"] > 0 and ss["edges_per_s"] > 0 and ss["local_rows"] > 0 and ss["layers"] == 3


def test_bench_partitioned_path_through_the_nccl_backend_and_the_native_exchange():
    """bench.py's N > 1 code path forced at world size 1 (SPEX_BENCH_FORCE_PARTITIONED=1) through the REAL `nccl` backend: the
    row partition, torch.distributed's RCCL collectives, the selection among the exchanges — where the library's own
    communicator (spex_comm_*: "native" and "native-p2p") is built, verified against the torch.distributed table and timed —
    and the strong-scaling leg.  With one rank nothing crosses a link, but every call of the driver's multi-GPU run is made."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, SPEX_BENCH_FORCE_PARTITIONED="1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "20", "--warmup", "5", "--hbm-log2-nodes", "17"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    ag = out["config"]["allgather"]
    assert "error" not in ag["native"], ag
    assert ag["native"]["native_equal"] is True and ag["native"]["native-p2p_equal"] is True, ag
    assert ag["native"]["propagate_ms_native"] > 0 and ag["used"] in ("collective", "peer", "native", "native-p2p")
    ss = out["extra"]["strong_scaling_hbm_graph"]
    assert "error" not in ss and ss["propagation_ms"] > 0, ss
    assert out["value"] > 0 and out["n_gpus"] == 1


def test_bench_multi_rank_rehearsal_verifies_the_native_exchange_with_real_data(tmp_path):
    """The same rehearsal with libspexhip bound to the functional shared-memory stand-in for RCCL (tests/stubs/rccl_shm_stub.c through
    SPEX_RCCL_LIB): bench.py's start-up selection then really builds the native communicator on both ranks, checks that "native" and
    "native-p2p" gather the very table torch.distributed's all-gather does — with data moving between the two processes — and times
    them; the code the driver's multi-GPU run executes first thing.  (The stand-in is stream-synchronous and slow, so it is not
    expected to be the schedule kept; RCCL itself is not involved.)"""
    import json
    import subprocess
    import sys
    so = str(tmp_path / "librccl_shm_stub.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(REPO, "tests", "stubs", "rccl_shm_stub.c"), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", so],
                   check=True)
    env = dict(os.environ, SPEX_BENCH_SHARE_GPU="1", SPEX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SPEX_RCCL_LIB=so,
               SPEX_ALLGATHER="collective")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3",
           "--no-strong-scaling"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    nat = out["config"]["allgather"]["native"]
    assert "error" not in nat, nat
    assert nat["native_equal"] is True and nat["native-p2p_equal"] is True, nat
    assert nat["propagate_ms_native"] > 0 and nat["propagate_ms_native-p2p"] > 0
    assert out["n_gpus"] == 2 and out["value"] > 0


def test_bench_single_gpu_line_is_well_formed():
    """bench.py as the driver runs it at N = 1 (here with the auxiliary measurements and the 2^24-node graph switched off, a
    short CPU-baseline leg left on): ONE JSON line carrying the contract's keys, `roofline` and `cpu_baseline` objects with
    consistent figures."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "20", "--warmup", "5", "--no-hbm-roofline", "--no-standalone"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5 and out["unit"] == "edges/s" and out["dtype"] == "f32"
    assert out["higher_is_better"] is True and out["vs_baseline"] is None and "workload" in out["config"]
    assert abs(out["value"] - 3 * 418608 * 20 / (out["ms_per_step"] * 20 * 1e-3)) <= 1e-6 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["launches_timed"] > 0
    # cache-resident workload: the algorithmic rate is reported apart (it may exceed the HBM peak); `frac` is L2-miss traffic (or
    # compulsory bytes) / time / peak and can never exceed 1
    assert abs(rf["cache_algorithmic_frac"] - rf["achieved"] / rf["peak"]) <= 1e-9
    assert 0.0 < rf["frac"] <= 1.0 and "frac_basis" in rf
    if rf.get("traffic"):
        assert abs(rf["frac"] - rf["traffic"] / (rf["avg_launch_us"] * 1e-6) / 1e9 / rf["peak"]) <= 1e-9
    tr = out["extra"]["timed_region"]
    assert 0.0 < tr["stream_events_ms"] <= tr["wall_ms"] * 1.05 and abs(tr["wall_ms"] - out["ms_per_step"] * 20) <= 1e-6 * tr["wall_ms"]
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * rf["achieved"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "edges/s" and "sample" in cb
    assert out["value"] > 20 * cb["value"]                  # (a sanity bound, not a target)


def _peer_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SPEX_PEER_CHECK="1")
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
    E0_local = torch.from_numpy(E0[P.r0:P.r1].copy()).to(dev)
    g_local = torch.from_numpy(E0[::-1].copy()[P.r0:P.r1].copy()).to(dev)
    ref_lo, ref_g = P.propagate(E0_local).clone(), P.propagate_bwd(g_local).clone()
    P.set_allgather("peer")
    outs = []
    for _ in range(5):                                   # several rounds: the two alternating buffers both get used
        outs.append((P.propagate(E0_local).clone(), P.propagate_bwd(g_local).clone()))
    same = all(torch.equal(a, ref_lo) and torch.equal(b, ref_g) for a, b in outs)
    P.set_allgather("collective")
    same = same and torch.equal(P.propagate(E0_local), ref_lo)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), same=same, calls=P.peer.calls, checked=P.peer.checked)
    dist.barrier()
    del P
    dist.destroy_process_group()


def test_peer_write_allgather_matches_the_collective(tmp_path):
    """The direct peer-write all-gather (IPC-mapped peer buffers, one copy per peer, a one-element all-reduce as the
    barrier, two alternating buffers) against torch.distributed's all-gather: identical propagated tables and gradients,
    forward and backward, over repeated calls; the barrier's token carries the ranks' call counters and every call checks that
    they met in the same call (a rank can never write two calls ahead of a peer's reads).  Two ranks share the test box's one GPU, so this exercises the handle
    exchange, the slot arithmetic and the ordering — not the xGMI links."""
    world = 2
    mp.spawn(_peer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert bool(d["same"]) and int(d["calls"]) == 5 * 6
        assert int(d["checked"]) == 5 * 6          # every barrier saw every rank in the SAME call (SPEX_PEER_CHECK: no rank a call ahead)


# ---------------------------------------------------------------------------------------------- collectives behind the C ABI
def _native_comm_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.dist import NativeComm, PartitionedLightGCN, PartitionedStepper
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    from spex_amd.trainer import LightGCNStepper
    tr = load_epinion2()["train"]
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    uw, iw = epinion2_tables(3186, 12407)
    E0 = torch.from_numpy(np.concatenate([uw, iw])).to(dev)
    rng = np.random.default_rng(3)
    batches = [(torch.from_numpy(rng.integers(0, 3185, 256)).to(dev), torch.from_numpy(rng.integers(0, 12407, 256)).to(dev),
                torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32)).to(dev)) for _ in range(4)]
    for b in batches:                                                    # repeated rows inside a batch
        b[0][:8] = b[0][0]
    factory = lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev)
    out = {}
    # raw collectives of the library's own communicator (RCCL bound inside libspexhip), world size 1
    comm = NativeComm(0, 1, dev)
    send = torch.randn(100, 64, device=dev)
    recv = torch.zeros(128, 64, device=dev)
    import ctypes
    comm.allgather_rows(send, recv, 128, (ctypes.c_int32 * 1)(100))
    buf = torch.randn(1000, device=dev)
    want = buf.clone()
    comm.allreduce_sum(buf)
    torch.cuda.synchronize()
    out["raw_ok"] = bool(torch.equal(recv[:100], send) and float(recv[100:].abs().sum()) == 0.0 and torch.equal(buf, want))
    comm.close()
    # ... and once more WITHOUT the world-size-1 shortcut (SPEX_COMM_NO_SHORTCUT, read when the communicator is created): RCCL accepts a
    # one-rank communicator, so ncclAllGather, ncclAllReduce and an (empty) ncclGroupStart / ncclGroupEnd pair really execute here.
    # What still has never executed anywhere: ncclSend / ncclRecv and any collective between two ranks.
    os.environ["SPEX_COMM_NO_SHORTCUT"] = "1"
    comm = NativeComm(0, 1, dev)
    del os.environ["SPEX_COMM_NO_SHORTCUT"]
    send2, recv2 = torch.randn(128, 64, device=dev), torch.zeros(128, 64, device=dev)
    comm.allgather_rows(send2, recv2, 128, None)                            # equal shards: ncclAllGather
    recv3 = torch.zeros(128, 64, device=dev)
    comm.allgather_rows(send, recv3, 128, (ctypes.c_int32 * 1)(100))        # real rows: own-slot copy + an empty group
    buf2 = buf.clone()
    comm.allreduce_sum(buf2)                                                # ncclAllReduce, in place, one rank: unchanged
    torch.cuda.synchronize()
    out["rccl_one_rank_ok"] = bool(torch.equal(recv2, send2) and torch.equal(recv3[:100], send) and float(recv3[100:].abs().sum()) == 0.0
                                   and torch.equal(buf2, want))
    comm.close()
    # the partitioned step three ways: collectives from Python (reference path of rounds 1-2), one native call, native deterministic
    results = {}
    for mode, det in (("collective", False), ("native", False), ("native-p2p", True), ("native-p2p", True)):
        P = PartitionedLightGCN(*csr, 3186, 3, 64, 0, 1, factory, dev, allgather=mode)
        st = PartitionedStepper(P, E0.clone(), lr=1e-3)
        acc = torch.zeros(1, device=dev)
        for bu, bi, by in batches:
            st.step_bce(bu, bi, by, loss_acc=acc, deterministic=det)
        torch.cuda.synchronize()
        results.setdefault((mode, det), []).append((st.E0.clone(), acc.clone(), P.propagate(st.E0).clone() if mode == "collective" else None))
        if mode != "collective":
            assert st.t == 4 and st._desc is not None
    single = LightGCNStepper(SpexGraph(*csr, device=dev), E0.clone(), 3186, n_layers=3, lr=1e-3, deterministic=True)
    acc1 = torch.zeros(1, device=dev)
    for bu, bi, by in batches:
        single.step_bce(bu, bi, by, loss_acc=acc1, batch_rows_only=True)
    ref_E0, ref_acc, _ = results[("collective", False)][0]
    nat_E0, nat_acc, _ = results[("native", False)][0]
    d1, d2 = results[("native-p2p", True)]
    out["native_vs_python"] = float((nat_E0 - ref_E0).abs().max() / ref_E0.abs().max())
    out["native_loss"] = abs(nat_acc.item() - ref_acc.item()) / abs(ref_acc.item())
    out["det_repeats"] = bool(torch.equal(d1[0], d2[0]) and torch.equal(d1[1], d2[1]))
    out["det_vs_single_device_det"] = float((d1[0] - single.E0).abs().max() / single.E0.abs().max())
    out["det_loss_vs_single"] = abs(d1[1].item() - acc1.item()) / abs(acc1.item())
    np.savez(os.path.join(out_dir, "native.npz"), **out)


def test_native_communicator_and_one_call_partitioned_step(tmp_path):
    """The collectives behind the C ABI (spex_comm_*: RCCL bound inside libspexhip) and the row-partitioned training
    step as ONE native call (spex_partitioned_step_bce_f32), on the test box's one GPU at world size 1.  What this covers of RCCL:
    ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy, and — on a communicator created under SPEX_COMM_NO_SHORTCUT=1 — one-rank
    ncclAllGather / ncclAllReduce / an empty group.  At world size 1 the library otherwise takes a local-copy shortcut BEFORE any
    RCCL call, so the steps below exercise the launch sequence, not a collective; ncclSend / ncclRecv and every two-rank collective
    have never executed (multi-rank RCCL needs one GPU per rank).  Their call sequence for world 2 / 4 / 8 is pinned against a
    recording stand-in in tests/test_rccl_stub.py; bench.py --gpus N checks the native exchange against torch.distributed's at
    start-up on a real node.  Here: the raw entry points; four training steps through the native call against the same steps with
    the collectives issued from Python; the deterministic mode repeating bit for bit and agreeing with the single-device
    deterministic stepper."""
    mp.spawn(_native_comm_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    d = np.load(tmp_path / "native.npz")
    assert bool(d["raw_ok"]) and bool(d["rccl_one_rank_ok"])
    # (the loss accumulator: an fp32 sum of 4 x 256 per-sample losses added by float atomics in arrival order on either side)
    assert float(d["native_vs_python"]) <= 2e-6 and float(d["native_loss"]) <= 3e-6, dict(d)
    assert bool(d["det_repeats"])
    assert float(d["det_vs_single_device_det"]) <= 5e-6 and float(d["det_loss_vs_single"]) <= 2e-6, dict(d)
