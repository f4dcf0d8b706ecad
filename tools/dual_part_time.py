"""Dual-task step on Epinion2, B = 256, 15 paths: DualTaskStepper (single-GPU one-call step) against PartitionedDualTaskStepper at
world size 1 (the row-partitioned one-call step: the same arithmetic through the partition's schedule — exchanges are local copies
here).  us per step by HIP events."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spex_amd", "dropin"))
from spex_amd.datasets import load_epinion2
from spex_amd.dist_dual import PartitionedDualTask, PartitionedDualTaskStepper
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import DualTaskStepper
import utility1.model_expert_s as mex
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
n_u, n_i, L = 3185, 12407, 3
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], n_u, n_i)
g = SpexGraph(*csr, device=dev)


class _DS:
    n_users, m_items = n_u, n_i
    getSparseGraph = staticmethod(lambda: g)


dargs = argparse.Namespace(hiddenSize=64, batchSize=100, nonhybrid=False, nb_heads=3, recdim=64, layer=L, keepprob=0.6, A_split=False, dropout=0)
rng = np.random.default_rng(13)
B, T, P_LEN = 256, int(os.environ.get("DUAL_PART_PATHS", "15")), 6            # DUAL_PART_PATHS=0: the rec branch alone
ub = torch.from_numpy(rng.integers(0, n_u, B)).to(dev); ib = torch.from_numpy(rng.integers(0, n_i, B)).to(dev)
yb = torch.from_numpy((rng.random(B) < 1 / 6).astype(np.float32)).to(dev)
plen = rng.integers(2, P_LEN + 1, T)
seq = np.full((T, P_LEN), n_u, dtype=np.int64)
for r, l in enumerate(plen):
    seq[r, :l] = rng.choice(n_u, size=l, replace=False)
seq_d, len_d = torch.from_numpy(seq).to(dev), torch.from_numpy(plen.astype(np.int64)).to(dev)
tgt = torch.from_numpy(rng.integers(0, n_u, T)).to(dev)
if T == 0:
    seq_d = len_d = tgt = None


def timed(fn, n=500, reps=3):
    """(best of `reps` runs of n steps by HIP events, the host's enqueue time per step in the same run), us"""
    import gc, time
    for _ in range(30): fn()
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()                 # (a full collection is ~40 ms: it would land inside one of the timed runs)
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


torch.manual_seed(0)
net = mex.LightGCN(dargs, _DS).to(dev)
st = DualTaskStepper(net, path_capacity=max(T, 1), path_len=P_LEN, lr=1e-3)
print("DualTaskStepper                                     : %.1f us (host enqueue %.1f)" % timed(lambda: st.step(ub, ib, yb, seq_d, len_d, tgt)), flush=True)
for det, fast, tag in ((False, True, "fast path      "), (False, False, "launch by launch"), (True, True, "deterministic   ")):
    torch.manual_seed(0)
    core = mex.LightGCN(dargs, _DS).to(dev)
    model = PartitionedDualTask(core, csr, 0, 1, dev)
    pst = PartitionedDualTaskStepper(model, path_capacity=max(T, 1), path_len=P_LEN, lr=1e-3, deterministic=det, fast=fast)
    pos = pst.positions(ub, ib)
    print("PartitionedDualTaskStepper, world 1, %s: %.1f us (host enqueue %.1f)" % ((tag,) + timed(lambda: pst.step(ub, ib, yb, seq_d, len_d, tgt, pos=pos))), flush=True)
