"""Exact LightGCN training step on Epinion2, B = 256, L = 3: LightGCNStepper (single-GPU one-call step) against PartitionedStepper at
world size 1 — the row-partitioned one-call step (spex_partitioned_step_bce_f32): fast path, launch-by-launch schedule, deterministic
mode.  Exchanges are the local-copy shortcut here (in place: nothing to copy but E^0).  us per step by HIP events + host enqueue time."""
import gc, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spex_amd.datasets import epinion2_tables, load_epinion2
from spex_amd.dist import PartitionedLightGCN, PartitionedStepper
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
n_u, n_i, L = 3185, 12407, 3
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], n_u, n_i)
uw, iw = epinion2_tables(n_u + 1, n_i)
E0 = torch.from_numpy(np.concatenate([uw, iw])).to(dev)
rng = np.random.default_rng(13)
B = 256
ub = torch.from_numpy(rng.integers(0, n_u, B)).to(dev); ib = torch.from_numpy(rng.integers(0, n_i, B)).to(dev)
yb = torch.from_numpy((rng.random(B) < 1 / 6).astype(np.float32)).to(dev)
acc = torch.zeros(1, device=dev)


def timed(fn, n=500, reps=3):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()
    best = (1e30, 0.0)
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n): fn()
        e1.record()
        host = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        best = min(best, (e0.elapsed_time(e1) / n * 1e3, host))
    return best


st = LightGCNStepper(SpexGraph(*csr, device=dev), E0.clone(), n_u + 1, n_layers=L, lr=1e-3)
print("LightGCNStepper (one-call step)                 : %.1f us (host enqueue %.1f)" % timed(lambda: st.step_bce(ub, ib, yb, loss_acc=acc, batch_rows_only=True)), flush=True)
for det, fast, tag in ((False, True, "fast path       "), (False, False, "launch by launch"), (True, True, "deterministic   ")):
    P = PartitionedLightGCN(*csr, n_u + 1, L, 64, 0, 1, lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev, allgather="native-p2p")
    pst = PartitionedStepper(P, E0.clone(), lr=1e-3, fast=fast)
    pos = pst.positions(ub, ib)
    print("PartitionedStepper, world 1, %s   : %.1f us (host enqueue %.1f)"
          % ((tag,) + timed(lambda: pst.step_bce(ub, ib, yb, pos=pos, loss_acc=acc, deterministic=det))), flush=True)
    P.native.close()

# ---- the same under the reference's recommended edge dropout (--dropout 1 --keepprob 0.3): a fresh sampled mask per step
from spex_amd.graph import csr_transpose
from spex_amd.trainer import edge_dropout_mask
n = len(csr[0]) - 1
t_rp, t_c, t_v, t_e = csr_transpose(*csr, n)
g1 = SpexGraph(*csr, device=dev)
st = LightGCNStepper(g1, E0.clone(), n_u + 1, n_layers=L, lr=1e-3, graph_t=SpexGraph(t_rp, t_c, t_v, n_cols=n, edge_id=t_e, device=dev))
k = {"k": 0}


def single_masked():
    k["k"] += 1
    mask = edge_dropout_mask(None, 0.3, "philox", 7, k["k"])
    st.graph.set_edge_mask(*mask); st.graph_t.set_edge_mask(*mask)
    st.step_bce(ub, ib, yb, loss_acc=acc, batch_rows_only=True)


print("edge dropout 0.3: LightGCNStepper (one-call step)     : %.1f us (host enqueue %.1f)" % timed(single_masked), flush=True)
efactory = lambda r, c, v, n_cols, edge_id=None: SpexGraph(r, c, v, n_cols=n_cols, edge_id=edge_id, device=dev)
for fast in (True, False):
    P = PartitionedLightGCN(*csr, n_u + 1, L, 64, 0, 1, efactory, dev, allgather="native-p2p", edge_ids=True)
    pst = PartitionedStepper(P, E0.clone(), lr=1e-3, fast=fast)
    pos = pst.positions(ub, ib)

    def part_masked():
        k["k"] += 1
        P.set_edge_mask(*edge_dropout_mask(None, 0.3, "philox", 7, k["k"]))
        pst.step_bce(ub, ib, yb, pos=pos, loss_acc=acc)
    print("edge dropout 0.3: PartitionedStepper, world 1, %s: %.1f us (host enqueue %.1f)"
          % ((("fast path       " if fast else "launch by launch"),) + timed(part_masked)), flush=True)
    P.native.close()
