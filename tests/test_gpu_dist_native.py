"""The library's NATIVE multi-rank paths (spex_amd/csrc/comm.hip: spex_comm_*, spex_partitioned_step_bce_f32,
spex_partitioned_dual_task_step_f32) executed for world > 1 on the one-GPU test box.

RCCL refuses two ranks on one device, so the ranks here — separate processes sharing cuda:0 — bind libspexhip to
tests/stubs/rccl_shm_stub.c through SPEX_RCCL_LIB: a functional stand-in that really moves the data between the processes through
host shared memory (stream-synchronous, deterministic) and refuses mismatched counts.  This runs every line of comm.hip's
multi-rank code with real data — the send / recv slot arithmetic included — but says nothing about RCCL or xGMI: SURVEY 8e stays
"unmeasured on hardware".  The schedule's reference analogue is the serial --A_split loop (utility1/model.py:84-89,
dataloader.py:167-177); the dual-task step is main_auto_expert_s.py:53-91."""
import os
import socket
import subprocess

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build_shm_stub(tmp_path):
    so = str(tmp_path / "librccl_shm_stub.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(HERE, "stubs", "rccl_shm_stub.c"), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", so], check=True)
    return so


def _setup(rank, world, port, stub):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "spex_amd", "dropin"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if stub:
        os.environ["SPEX_RCCL_LIB"] = stub                 # (before the library binds RCCL: once per process)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # the bootstrap only: hands the 128-byte id to every rank
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    return dev


# ---------------------------------------------------------------------------------------------- LightGCN: exchange + one-call step
def _light_worker(rank, world, port, out_dir, stub):
    dev = _setup(rank, world, port, stub)
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.dist import PartitionedLightGCN, PartitionedStepper
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    tr = load_epinion2()["train"]
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    uw, iw = epinion2_tables(3186, 12407)
    E0 = np.concatenate([uw, iw])
    factory = lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev)
    n = len(csr[0]) - 1
    bounds = np.linspace(0, n, world + 1).astype(np.int64)
    bounds[1] -= 37                                                     # uneven shards: the slots' padding tails are exercised
    rng = np.random.default_rng(3)
    batches = [(torch.from_numpy(rng.integers(0, 3185, 256)).to(dev), torch.from_numpy(rng.integers(0, 12407, 256)).to(dev),
                torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32)).to(dev)) for _ in range(3)]
    out = {}
    res = {}
    for mode, det in (("collective", False), ("native", False), ("native-p2p", False), ("native-p2p", True), ("native-p2p", True)):
        P = PartitionedLightGCN(*csr, 3186, 3, 64, rank, world, factory, dev, bounds=bounds, allgather=mode)
        E0_local = torch.from_numpy(E0[P.r0:P.r1].copy()).to(dev)
        lo = P.propagate(E0_local).clone()
        gr = P.propagate_bwd(torch.from_numpy(E0[::-1].copy()[P.r0:P.r1].copy()).to(dev)).clone()
        st = PartitionedStepper(P, E0_local.clone(), lr=1e-3)
        acc = torch.zeros(1, device=dev)
        for bu, bi, by in batches:
            st.step_bce(bu, bi, by, loss_acc=acc, deterministic=det)
        torch.cuda.synchronize()
        res.setdefault((mode, det), []).append((lo, gr, st.E0.clone(), acc.clone()))
        if mode != "collective":
            assert st._desc is not None and st.t == 3                   # the one-call native step ran
            P.native.close()
        out["r0"], out["r1"] = P.r0, P.r1
    ref = res[("collective", False)][0]
    for key in (("native", False), ("native-p2p", False)):
        got = res[key][0]
        out["%s_prop_equal" % key[0]] = bool(torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]))
        out["%s_step" % key[0]] = float((got[2] - ref[2]).abs().max() / ref[2].abs().max())
        out["%s_loss" % key[0]] = abs(got[3].item() - ref[3].item()) / abs(ref[3].item())
    # ---- the reference's recommended configuration on the partition (`--dropout 1 --keepprob 0.3`, README.md:119-123): blocks with
    #      GLOBAL edge ids, the sampled mask keyed by them — every rank drops the same edges of A and of A^T as one device does
    from spex_amd.trainer import edge_dropout_mask
    efactory = lambda r, c, v, n_cols, edge_id=None: SpexGraph(r, c, v, n_cols=n_cols, edge_id=edge_id, device=dev)
    P = PartitionedLightGCN(*csr, 3186, 3, 64, rank, world, efactory, dev, bounds=bounds, allgather="native-p2p", edge_ids=True)
    st = PartitionedStepper(P, torch.from_numpy(E0[P.r0:P.r1].copy()).to(dev), lr=1e-3)
    acc = torch.zeros(1, device=dev)
    for k, (bu, bi, by) in enumerate(batches):
        P.set_edge_mask(*edge_dropout_mask(P.graph, 0.3, "philox", 5, k + 1))
        st.step_bce(bu, bi, by, loss_acc=acc)
    torch.cuda.synchronize()
    out["dropout_trained"], out["dropout_loss"] = st.E0.cpu().numpy(), acc.item()
    P.native.close()
    d1, d2 = res[("native-p2p", True)]
    out["det_repeats"] = bool(torch.equal(d1[2], d2[2]) and torch.equal(d1[3], d2[3]))
    out["det_step"] = float((d1[2] - ref[2]).abs().max() / ref[2].abs().max())
    out["trained"], out["lo"] = d1[2].cpu().numpy(), ref[0].cpu().numpy()
    np.savez(os.path.join(out_dir, f"light{world}_{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_native_exchange_and_one_call_step_with_real_data_between_ranks(tmp_path, world):
    """world 2 and 3 (uneven shards) on Epinion2: both native exchange forms give the very tables torch.distributed's all-gather
    gives (forward and backward propagation bit-identical), three one-call native training steps equal the Python-issued steps
    (<= 2e-6), the deterministic mode repeats bit for bit; and the ranks' rows together are the single-device result.  And under edge
    dropout (`--dropout 1 --keepprob 0.3`, utility1/model.py:46-64): three one-call steps with a fresh sampled mask per step on blocks
    that carry GLOBAL edge ids equal the same steps under the same masks on one device (5e-6)."""
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    stub = _build_shm_stub(tmp_path)
    mp.spawn(_light_worker, args=(world, _free_port(), str(tmp_path), stub), nprocs=world, join=True)
    tr = load_epinion2()["train"]
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
    uw, iw = epinion2_tables(3186, 12407)
    ref = SpexGraph(*csr).propagate(torch.from_numpy(np.concatenate([uw, iw])).cuda(), 3).cpu().numpy()
    # the same three steps under the same sampled masks on ONE device (the launch-by-launch stepper: masked whole-graph products)
    from spex_amd.graph import csr_transpose
    from spex_amd.trainer import LightGCNStepper, edge_dropout_mask
    g1 = SpexGraph(*csr)
    t_rp, t_c, t_v, t_e = csr_transpose(*csr, len(csr[0]) - 1)
    single = LightGCNStepper(g1, torch.from_numpy(np.concatenate([uw, iw])).cuda(), 3186, n_layers=3, lr=1e-3,
                             graph_t=SpexGraph(t_rp, t_c, t_v, n_cols=len(csr[0]) - 1, edge_id=t_e))
    rng = np.random.default_rng(3)
    acc1 = torch.zeros(1, device="cuda")
    for k in range(3):
        bu, bi = torch.from_numpy(rng.integers(0, 3185, 256)).cuda(), torch.from_numpy(rng.integers(0, 12407, 256)).cuda()
        by = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32)).cuda()
        mask = edge_dropout_mask(g1, 0.3, "philox", 5, k + 1)
        single.graph.set_edge_mask(*mask); single.graph_t.set_edge_mask(*mask)
        single.step_bce(bu, bi, by, loss_acc=acc1)
    want_drop = single.E0.cpu().numpy()
    got_drop = np.zeros_like(want_drop)
    lo = np.zeros_like(ref)
    for r in range(world):
        d = np.load(tmp_path / f"light{world}_{r}.npz")
        got_drop[int(d["r0"]):int(d["r1"])] = d["dropout_trained"]
        assert abs(float(d["dropout_loss"]) - acc1.item()) <= 2e-6 * abs(acc1.item()), (r, float(d["dropout_loss"]), acc1.item())
        assert bool(d["native_prop_equal"]) and bool(d["native-p2p_prop_equal"]), r
        assert float(d["native_step"]) <= 2e-6 and float(d["native-p2p_step"]) <= 2e-6, dict(d)
        assert float(d["native_loss"]) <= 3e-6 and float(d["native-p2p_loss"]) <= 3e-6   # (fp32 sums of per-sample losses in arrival order)
        assert bool(d["det_repeats"]) and float(d["det_step"]) <= 5e-6
        lo[int(d["r0"]):int(d["r1"])] = d["lo"]
    assert np.array_equal(lo, ref)
    assert np.abs(got_drop - want_drop).max() <= 5e-6 * np.abs(want_drop).max()
    assert np.abs(want_drop - np.concatenate([uw, iw])).max() > 1e-4                                   # (the masked steps did move the table)


# ---------------------------------------------------------------------------------------------- BASELINE config 5: the one-call dual-task step
def _dual_native_worker(rank, world, port, out_dir, data_root, n_steps, stub, det, L=3):
    import random
    from collections import defaultdict
    dev = _setup(rank, world, port, stub)
    from torch.utils.data import DataLoader
    import lg_parser
    import utility1.dataloader as dl
    import utility1.model_expert_s as mex
    import utility1.utils as utils
    from utility2.utils import Data
    from spex_amd.dist_dual import PartitionedDualTask, PartitionedDualTaskStepper
    t = np.load(os.path.join(GOLDEN, "trust_epinion2_paths.npz"))
    raw_train = ([r[:l].tolist() for r, l in zip(t["train_paths"].astype(np.int64), t["train_len"])],
                 t["train_targets"].astype(np.int64).tolist())
    args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", data_root, "--layer", str(L)])
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
    st = PartitionedDualTaskStepper(model, path_capacity=cap, path_len=train2.len_max, lr=args.lr, deterministic=det,
                                    exchange="native-p2p" if rank_world_p2p(world) else "native")
    loader.dataset.ng_sample()
    core.train()
    l1s, l2s, n_paths = [], [], []
    prev = np.zeros(2)
    for step, (user, item, label) in enumerate(loader):
        if step == n_steps:
            break
        chosen = []
        for u in set(user.numpy().tolist()):
            chosen.extend(by_user[u])
        if len(chosen) > cap:
            chosen = random.sample(chosen, cap)
        inputs, mask, targets = train2.get_slice(np.array(chosen, dtype=int))
        seq = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.int64)).to(dev)
        seq_l = torch.from_numpy(np.asarray(mask).sum(1).astype(np.int64)).to(dev)
        tgt = torch.from_numpy(np.asarray(targets).astype(np.int64)).to(dev)
        st.step(user.to(dev), item.to(dev), label.to(dev).float(), seq, seq_l, tgt)
        cur = st.loss_acc.cpu().numpy().astype(np.float64)
        l1s.append(cur[0] - prev[0]); l2s.append(cur[1] - prev[1]); n_paths.append(len(chosen))
        prev = cur
    np.savez(os.path.join(out_dir, f"dual{world}_{int(det)}_{L}_{rank}.npz"), loss1=np.asarray(l1s), loss2=np.asarray(l2s), n_paths=np.asarray(n_paths),
             r0=model.P.r0, r1=model.P.r1, table=model.E0_local.detach().cpu().numpy(),
             task_weights=model.task_weights.detach().cpu().numpy(), att_exp1=core.att_exp1.detach().cpu().numpy(),
             w=core.w.detach().cpu().numpy())
    dist.barrier()
    model.P.native.close()
    dist.destroy_process_group()


def rank_world_p2p(world):
    return world != 3                                       # world 3 takes the equal-shard form, the others the send / recv form


@pytest.mark.parametrize("world,det,L", [(1, False, 3), (2, False, 3), (2, True, 3), (3, False, 3), (2, False, 2), (2, False, 4)])
def test_one_call_partitioned_dual_task_step_reproduces_the_reference_losses(tmp_path, golden, world, det, L):
    """BASELINE config 5 row-partitioned, every step ONE native call (spex_partitioned_dual_task_step_f32): the first 16 training
    steps of main_auto_expert_s.py on Epinion2 + the reference-minted trust paths reproduce the REFERENCE's per-step losses of both
    tasks (golden G13) at world 1 (real RCCL communicator, local-copy shortcut), 2 and 3 (shm stand-in: real data between the
    ranks); every rank ends with identical replicated parameters (the gate gradients need no collective: the batch's rows are
    gated redundantly), and the ranks' table rows tile the whole table.  L = 2 / 4: `main_auto_expert_s.py --layer 2` / `--layer 4`
    (goldens dual_epinion2_L{2,4}_epochs) — the fast path's other two schedules (the mean's share of the one pull product added by the
    Adam pass; the running-sum forward and add-form products)."""
    from spex_amd.datasets import materialise_epinion2
    g = golden("dual_epinion2_epochs" if L == 3 else f"dual_epinion2_L{L}_epochs")
    root = materialise_epinion2(str(tmp_path / "data"))
    stub = _build_shm_stub(tmp_path) if world > 1 else ""
    n_steps = len(g["loss1_first"])
    mp.spawn(_dual_native_worker, args=(world, _free_port(), str(tmp_path), root, n_steps, stub, det, L), nprocs=world, join=True)
    d = [np.load(tmp_path / f"dual{world}_{int(det)}_{L}_{r}.npz") for r in range(world)]
    for r in range(world):
        assert np.array_equal(d[r]["n_paths"], g["n_paths"][:n_steps].astype(int))
        assert np.abs(d[r]["loss1"] - g["loss1_first"]).max() <= 2e-5, (r, d[r]["loss1"], g["loss1_first"])
        assert (np.abs(d[r]["loss2"] - g["loss2_first"]) <= 1e-4 * g["loss2_first"]).all(), (r, d[r]["loss2"], g["loss2_first"])
    for r in range(1, world):
        for k in ("task_weights", "att_exp1", "w"):
            assert np.array_equal(d[0][k], d[r][k]) or np.abs(d[0][k] - d[r][k]).max() <= 1e-7 * max(1.0, np.abs(d[0][k]).max()), k
        assert int(d[r - 1]["r1"]) == int(d[r]["r0"])
    assert int(d[0]["r0"]) == 0 and int(d[-1]["r1"]) == 3186 + 12407


@pytest.mark.parametrize("L", [2, 3, 4])
def test_partitioned_dual_fast_path_equals_the_launch_by_launch_schedule(L):
    """spex_partitioned_dual_task_step_f32's fast path (plain forward layers, the last one at the batch's rows on their owners, one
    launch for gate + scores + the gate's backward, the backward's first product in push form without an exchange, the layer-mean
    share of the last product left to the Adam pass — for L == 3 the all-plain backward) against the launch-by-launch schedule of
    the same call (fast=False): the same arithmetic in another order of float additions, so after four steps every parameter of the
    arena, both Adam moments and both loss sums agree to rounding.  World size 1 (real RCCL communicator, local-copy shortcut);
    L = 2 / 3 / 4 take the three branches of the schedule (model_expert_s.py:95-126,154-168; main_auto_expert_s.py:53-91)."""
    import argparse
    import sys
    sys.path.insert(0, os.path.join(REPO, "spex_amd", "dropin"))
    import utility1.model_expert_s as mex
    from spex_amd.datasets import load_epinion2
    from spex_amd.dist_dual import PartitionedDualTask, PartitionedDualTaskStepper
    from spex_amd.graph import lightgcn_norm_adj
    dev = torch.device("cuda:0")
    tr = load_epinion2()["train"]
    n_u, n_i = 3185, 12407
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], n_u, n_i)

    class _DS:
        n_users, m_items = n_u, n_i
        getSparseGraph = staticmethod(lambda: None)
    dargs = argparse.Namespace(hiddenSize=64, batchSize=100, nonhybrid=False, nb_heads=3, recdim=64, layer=L, keepprob=0.6, A_split=False, dropout=0)
    rng = np.random.default_rng(17)
    B, T, P_LEN = 256, 9, 6
    batches = []
    for _ in range(4):
        ub = torch.from_numpy(np.r_[rng.integers(0, n_u, B - 8), np.full(8, 7)]).to(dev)        # a user named nine times in the batch
        ib = torch.from_numpy(rng.integers(0, n_i, B)).to(dev)
        yb = torch.from_numpy((rng.random(B) < 1 / 6).astype(np.float32)).to(dev)
        plen = rng.integers(2, P_LEN + 1, T)
        seq = np.full((T, P_LEN), n_u, dtype=np.int64)
        for r, l in enumerate(plen):
            seq[r, :l] = rng.choice(n_u, size=l, replace=False)
        batches.append((ub, ib, yb, torch.from_numpy(seq).to(dev), torch.from_numpy(plen.astype(np.int64)).to(dev),
                        torch.from_numpy(rng.integers(0, n_u, T)).to(dev)))
    out = {}
    for fast in (True, False):
        torch.manual_seed(0)
        core = mex.LightGCN(dargs, _DS).to(dev)
        model = PartitionedDualTask(core, csr, 0, 1, dev)
        st = PartitionedDualTaskStepper(model, path_capacity=T, path_len=P_LEN, lr=1e-3, fast=fast)
        for b in batches:
            st.step(*b)
        torch.cuda.synchronize()
        out[fast] = [x.detach().cpu().numpy().astype(np.float64) for x in (st.arena, st.m, st.v, st.loss_acc)]
        # what the step must leave all-zero for the next one
        assert float(st.g_prop.abs().sum()) == 0.0 and float(st.g_raw.abs().sum()) == 0.0 and float(st.g_small.abs().sum()) == 0.0
        if fast:
            assert float(model.P.table(2)[: model.P.n_local].abs().sum()) == 0.0                  # the push target (rank 0's own slot)
            assert float(st.grad_slots.abs().sum()) == 0.0                                       # the gate gradients' copies
        model.P.native.close()
    for a, b, name in zip(out[True], out[False], ("parameters", "m", "v", "loss sums")):
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        assert err <= 5e-6, (L, name, err)


@pytest.mark.parametrize("L", [2, 3, 4])
def test_partitioned_lightgcn_fast_path_equals_the_launch_by_launch_schedule(L):
    """spex_partitioned_step_bce_f32's fast path (last forward layer at the batch's rows on their owners, scores + owner-computes adds
    in one launch, the backward's first product in push form without an exchange, the mean's share of the last product added by the
    Adam pass; L == 3: the all-plain backward) against the launch-by-launch schedule of the same call and against the single-device
    one-call step (LightGCNStepper): three steps on Epinion2, world size 1, L = 2 / 3 / 4 (utility1/model.py:66-121, main_rec.py:32-37)."""
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.dist import PartitionedLightGCN, PartitionedStepper
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    from spex_amd.trainer import LightGCNStepper
    dev = torch.device("cuda:0")
    tr = load_epinion2()["train"]
    n_u, n_i = 3185, 12407
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], n_u, n_i)
    uw, iw = epinion2_tables(n_u + 1, n_i)
    E0 = torch.from_numpy(np.concatenate([uw, iw])).to(dev)
    rng = np.random.default_rng(23)
    B = 256
    batches = [(torch.from_numpy(np.r_[rng.integers(0, n_u, B - 6), np.full(6, 11)]).to(dev), torch.from_numpy(rng.integers(0, n_i, B)).to(dev),
                torch.from_numpy((rng.random(B) < 1 / 6).astype(np.float32)).to(dev)) for _ in range(3)]
    out = {}
    for fast in (True, False):
        P = PartitionedLightGCN(*csr, n_u + 1, L, 64, 0, 1, lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev,
                                allgather="native-p2p")
        st = PartitionedStepper(P, E0.clone(), lr=1e-3, fast=fast)
        acc = torch.zeros(1, device=dev)
        for u, i, y in batches:
            st.step_bce(u, i, y, loss_acc=acc)
        torch.cuda.synchronize()
        out[fast] = [x.cpu().numpy().astype(np.float64) for x in (st.E0, st.m, st.v, acc)]
        assert float(st.g_local.abs().sum()) == 0.0
        if fast:
            assert float(P.table(2)[: P.n_local].abs().sum()) == 0.0                       # the push target is left all-zero
        P.native.close()
    single = LightGCNStepper(SpexGraph(*csr, device=dev), E0.clone(), n_u + 1, n_layers=L, lr=1e-3)
    acc = torch.zeros(1, device=dev)
    for u, i, y in batches:
        single.step_bce(u, i, y, loss_acc=acc)
    torch.cuda.synchronize()
    out["single"] = [x.cpu().numpy().astype(np.float64) for x in (single.E0, single.m, single.v, acc)]
    for other in (False, "single"):
        for a, b, name in zip(out[True], out[other], ("table", "m", "v", "loss sum")):
            err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
            assert err <= 5e-6, (L, other, name, err)


@pytest.mark.parametrize("L", [2, 3, 4])
def test_partitioned_fast_path_on_a_non_symmetric_matrix_against_the_oracle(L, oracle):
    """The fast path's push structure is the transpose of the rank's block of A^T (PartitionedLightGCN.push_graph): on a NON-symmetric
    A with a hub row the gradient of one spex_partitioned_step_bce_f32 — read back from Adam's first moment, m = (1 - beta1) g — is
    the oracle's d loss / d E0 (forward on A, backward on A^T: autograd of utility1/model.py:83-97,111-121), on the fast path and
    on the launch-by-launch schedule alike (a push over the rows of A^T instead of A's would pass every symmetric test)."""
    from spex_amd.dist import PartitionedLightGCN, PartitionedStepper
    from spex_amd.graph import SpexGraph
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(31 + L)
    n, n_u, B = 2000, 800, 96
    deg = rng.integers(0, 40, n)
    deg[5], deg[900], deg[7] = 1500, 1100, 0
    rowptr = np.zeros(n + 1, np.int64)
    cols = []
    for r in range(n):
        cols.append(np.sort(rng.choice(n, int(deg[r]), replace=False)))
        rowptr[r + 1] = rowptr[r] + deg[r]
    col = np.concatenate(cols).astype(np.int32)
    val = (rng.random(len(col)).astype(np.float32) * 0.2 + 0.01)
    rowptr = rowptr.astype(np.int32)
    t_csr = oracle.csr_transpose(rowptr, col, val, n)
    E0 = (rng.normal(size=(n, 64)) * 0.3).astype(np.float32)
    u = np.r_[rng.integers(0, n_u, B - 3), [5, 5, 7]]                      # the hub row twice, the empty row once
    i = np.r_[rng.integers(0, n - n_u, B - 1), [900 - n_u]]
    y = (rng.random(B) < 0.3).astype(np.float32)
    _, loss, G = oracle.lightgcn_loss_and_grad(rowptr, col, val, E0, n_u, L, u, i, y, n_threads=4, t_csr=t_csr)
    for fast in (True, False):
        P = PartitionedLightGCN(rowptr, col, val, n_u, L, 64, 0, 1, lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev,
                                t_csr=t_csr, allgather="native")
        st = PartitionedStepper(P, torch.from_numpy(E0).to(dev), lr=1e-3, fast=fast)
        acc = torch.zeros(1, device=dev)
        st.step_bce(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(y).to(dev), loss_acc=acc)
        torch.cuda.synchronize()
        got = st.m.cpu().numpy().astype(np.float64) / (1.0 - 0.9)
        err = np.abs(got - G).max() / np.abs(G).max()
        assert err <= 2e-5, (L, fast, err)
        assert abs(acc.item() / B - float(loss)) <= 2e-6 * max(1.0, abs(float(loss))), (acc.item() / B, loss)
        P.native.close()


def test_partitioned_steps_under_edge_dropout_equal_the_one_gpu_steps():
    """Edge dropout on the row partition (the reference's recommended `--dropout 1 --keepprob 0.3`, README.md:119-123;
    utility1/model.py:46-64, model_expert_s.py:104-109), world size 1: blocks built with GLOBAL edge ids (edge_ids=True) take the
    step's mask on both handles AND on the push structure, and the one-call steps keep their fast path — the rows-only last layer and
    the push apply the handles' keep rule entry by entry, the other products are masked whole-block launches.  (a) spex_partitioned_step_bce_f32 with an INJECTED keep mask (the reference-stream form: an array indexed by edge id)
    and with the sampled one against LightGCNStepper under the same masks; (b) spex_partitioned_dual_task_step_f32 with sampled masks
    against DualTaskStepper under the same masks (the rec branch drops edges, the trust branch reads the raw table): losses and every
    parameter after three steps."""
    import argparse
    import sys
    sys.path.insert(0, os.path.join(REPO, "spex_amd", "dropin"))
    import utility1.model_expert_s as mex
    from spex_amd.datasets import epinion2_tables, load_epinion2
    from spex_amd.dist import PartitionedLightGCN, PartitionedStepper
    from spex_amd.dist_dual import PartitionedDualTask, PartitionedDualTaskStepper
    from spex_amd.graph import SpexGraph, csr_transpose, lightgcn_norm_adj
    from spex_amd.trainer import DualTaskStepper, LightGCNStepper, edge_dropout_mask
    dev = torch.device("cuda:0")
    tr = load_epinion2()["train"]
    n_u, n_i, L = 3185, 12407, 3
    csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], n_u, n_i)
    n, nnz = len(csr[0]) - 1, len(csr[1])
    uw, iw = epinion2_tables(n_u + 1, n_i)
    E0 = torch.from_numpy(np.concatenate([uw, iw])).to(dev)
    rng = np.random.default_rng(29)
    B = 256
    batches = [(torch.from_numpy(rng.integers(0, n_u, B)).to(dev), torch.from_numpy(rng.integers(0, n_i, B)).to(dev),
                torch.from_numpy((rng.random(B) < 1 / 6).astype(np.float32)).to(dev)) for _ in range(3)]
    keep = torch.from_numpy((rng.random(nnz) < 0.3).astype(np.uint8)).to(dev)
    masks = [(1, keep, 0.3, 0), edge_dropout_mask(None, 0.3, "philox", 9, 2), edge_dropout_mask(None, 0.3, "philox", 9, 3)]
    efactory = lambda r, c, v, n_cols, edge_id=None: SpexGraph(r, c, v, n_cols=n_cols, edge_id=edge_id, device=dev)
    t_rp, t_c, t_v, t_e = csr_transpose(*csr, n)
    # ---- (a) LightGCN
    P = PartitionedLightGCN(*csr, n_u + 1, L, 64, 0, 1, efactory, dev, allgather="native-p2p", edge_ids=True)
    assert P.graph_t is not P.graph
    part = PartitionedStepper(P, E0.clone(), lr=1e-3)
    single = LightGCNStepper(SpexGraph(*csr, device=dev), E0.clone(), n_u + 1, n_layers=L, lr=1e-3,
                             graph_t=SpexGraph(t_rp, t_c, t_v, n_cols=n, edge_id=t_e, device=dev))
    acc_p, acc_s = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    for (u, i, y), mask in zip(batches, masks):
        P.set_edge_mask(*mask)
        single.graph.set_edge_mask(*mask); single.graph_t.set_edge_mask(*mask)
        part.step_bce(u, i, y, loss_acc=acc_p)
        single.step_bce(u, i, y, loss_acc=acc_s)
    torch.cuda.synchronize()
    assert abs(acc_p.item() - acc_s.item()) <= 2e-6 * abs(acc_s.item())
    err = (part.E0 - single.E0).abs().max().item() / single.E0.abs().max().item()
    assert err <= 5e-6, err
    # the masked steps ran the FAST path: its push structure exists, carries global edge ids and the step's mask
    assert P._graph_push is not None and P._graph_push.mask_mode == 2 and P._graph_push._edge_id_host is not None
    slow = PartitionedStepper(P, E0.clone(), lr=1e-3, fast=False)                         # ... and equals the launch-by-launch schedule
    acc_l = torch.zeros(1, device=dev)
    for (u, i, y), mask in zip(batches, masks):
        P.set_edge_mask(*mask)
        slow.step_bce(u, i, y, loss_acc=acc_l)
    assert (slow.E0 - part.E0).abs().max().item() <= 5e-6 * part.E0.abs().max().item() and abs(acc_l.item() - acc_p.item()) <= 2e-6 * acc_p.item()
    unmasked = PartitionedStepper(PartitionedLightGCN(*csr, n_u + 1, L, 64, 0, 1, efactory, dev, allgather="native-p2p"), E0.clone(), lr=1e-3)
    for u, i, y in batches:
        unmasked.step_bce(u, i, y, loss_acc=torch.zeros(1, device=dev))
    assert (unmasked.E0 - part.E0).abs().max().item() > 1e-4                              # (the masks mattered)
    P.set_edge_mask(0)
    P.native.close(); unmasked.P.native.close()
    # ---- (b) the dual-task step
    g1 = SpexGraph(*csr, device=dev)

    class _DS:
        n_users, m_items = n_u, n_i
        getSparseGraph = staticmethod(lambda: g1)
    dargs = argparse.Namespace(hiddenSize=64, batchSize=100, nonhybrid=False, nb_heads=3, recdim=64, layer=L, keepprob=0.3, A_split=False, dropout=1)
    T, P_LEN = 9, 6
    plen = rng.integers(2, P_LEN + 1, T)
    seq = np.full((T, P_LEN), n_u, dtype=np.int64)
    for r, l in enumerate(plen):
        seq[r, :l] = rng.choice(n_u, size=l, replace=False)
    seq_d, len_d = torch.from_numpy(seq).to(dev), torch.from_numpy(plen.astype(np.int64)).to(dev)
    tgt = torch.from_numpy(rng.integers(0, n_u, T)).to(dev)
    torch.manual_seed(0)
    net = mex.LightGCN(dargs, _DS).to(dev)
    dsingle = DualTaskStepper(net, path_capacity=T, path_len=P_LEN, lr=1e-3)
    torch.manual_seed(0)
    core = mex.LightGCN(dargs, _DS).to(dev)
    model = PartitionedDualTask(core, csr, 0, 1, dev, edge_ids=True)
    dpart = PartitionedDualTaskStepper(model, path_capacity=T, path_len=P_LEN, lr=1e-3)
    for k, (u, i, y) in enumerate(batches):
        mask = edge_dropout_mask(None, 0.3, "philox", 21, k + 1)
        dsingle.set_edge_dropout(mask)
        model.P.set_edge_mask(*mask)
        dsingle.step(u, i, y, seq_d, len_d, tgt)
        dpart.step(u, i, y, seq_d, len_d, tgt)
    torch.cuda.synchronize()
    a, b = dsingle.loss_acc.cpu().numpy(), dpart.loss_acc.cpu().numpy()
    assert np.abs(a - b).max() <= 3e-6 * np.abs(a).max(), (a, b)
    want = torch.cat([net.embedding_user.weight, net.embedding_item.weight]).detach()
    assert (model.E0_local.detach() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    for (name, p_), (_, q_) in zip(net.named_parameters(), core.named_parameters()):
        if not name.startswith("embedding_"):
            # (Adam's first steps move every touched parameter by ~lr whatever the gradient's size: compare on that scale)
            assert (p_.detach() - q_.detach()).abs().max().item() <= 0.02 * 1e-3 * 3, name
    dsingle.set_edge_dropout(None)
    model.P.native.close()
