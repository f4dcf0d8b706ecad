"""The exact LightGCN step and the dual-task step on the Weibo-SHAPED graph (BASELINE configs 3 / 5: 6 812 users, 20 000 items,
~500 k stored entries, log-normal user activity, Zipf item popularity: hub rows far beyond 1 024 entries) — us per step by HIP
events; under `rocprofv3 --kernel-trace --stats` it gives the per-kernel picture for that shape."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spex_amd", "dropin"))
if os.environ.get("SPEX_LIB"):
    from spex_amd import _lib as _l
    _l.LIB_PATH = os.environ["SPEX_LIB"]          # A/B against another build of the library
from spex_amd.datasets import synthetic_interactions, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import DualTaskStepper, LightGCNStepper
import utility1.model_expert_s as mex
dev = torch.device("cuda:0")
n_u, n_i, L, D, B = 6812, 20000, 3, 64, 256
u, i = synthetic_interactions(n_u, n_i, 400000, seed=7, sigma=1.4)
csr = lightgcn_norm_adj(u.numpy(), i.numpy(), n_u, n_i)
deg = np.diff(csr[0])
print("weibo shape: nnz %d, max user row %d, max item row %d, rows > 1024 entries: %d" % (len(csr[1]), deg[:n_u + 1].max(), deg[n_u + 1:].max(), (deg > 1024).sum()))
g = SpexGraph(*csr, device=dev)
rng = np.random.default_rng(13)
E0 = torch.from_numpy(np.concatenate([xavier_uniform_np(n_u + 1, D, rng), xavier_uniform_np(n_i, D, rng)])).to(dev)
st = LightGCNStepper(g, E0.clone(), n_u + 1, n_layers=L, lr=1e-3)
ub = torch.from_numpy(rng.integers(0, n_u, B)).to(dev); ib = torch.from_numpy(rng.integers(0, n_i, B)).to(dev)
yb = torch.from_numpy((rng.random(B) < 1 / 6).astype(np.float32)).to(dev)
acc = torch.zeros(1, device=dev)
def timed(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("exact LightGCN step, uniform batch: %.1f us" % timed(lambda: st.step_bce(ub, ib, yb, loss_acc=acc, batch_rows_only=True)))
X = E0.clone(); Y = torch.empty_like(X)
print("plain SpMM launch: %.1f us" % timed(lambda: g.spmm(X, Y=Y)))
class _DS:
    n_users, m_items = n_u, n_i
    getSparseGraph = staticmethod(lambda: g)
dargs = argparse.Namespace(hiddenSize=64, batchSize=100, nonhybrid=False, nb_heads=3, recdim=64, layer=L, keepprob=0.6, A_split=False, dropout=0)
net = mex.LightGCN(dargs, _DS).to(dev)
T, P_LEN = 15, 6
dst = DualTaskStepper(net, path_capacity=T, path_len=P_LEN, lr=1e-3)
plen = rng.integers(2, P_LEN + 1, T)
seq = np.full((T, P_LEN), n_u, dtype=np.int64)
for r, l in enumerate(plen):
    seq[r, :l] = rng.choice(n_u, size=l, replace=False)
seq_d, len_d = torch.from_numpy(seq).to(dev), torch.from_numpy(plen.astype(np.int64)).to(dev)
tgt = torch.from_numpy(rng.integers(0, n_u, T)).to(dev)
print("dual-task step, 15 paths: %.1f us" % timed(lambda: dst.step(ub, ib, yb, seq_d, len_d, tgt)))
