"""Phase stamps of lightgcn_batch_kernel (debug build: make -C spex_amd/csrc FLAGS+=-DSPEX_STAMPS, a throw-away library).
Prints, for workgroup 8 / wave 0 of 20 launches inside the exact step, the time between stamps in us (wall_clock64, 10 ns)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd import _lib
from spex_amd.datasets import load_epinion2, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
rng = np.random.default_rng(0)
E0 = torch.from_numpy(np.concatenate([xavier_uniform_np(3186, 64, rng), xavier_uniform_np(12407, 64, rng)])).to(dev)
st = LightGCNStepper(SpexGraph(*csr, device=dev), E0, 3186, n_layers=3, lr=1e-3)
u = torch.randint(0, 3185, (256,), device=dev); i = torch.randint(0, 12407, (256,), device=dev)
y = (torch.rand(256, device=dev) < 1 / 6).float()
acc = torch.zeros(1, device=dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
names = ["idx+rowptr+prefetch issued", "segment sums (gathers)", "barrier 1", "combine + barrier 2", "dot + dg + row atomics", "push issue"]
rows = []
for k in range(30):
    st.step_bce(u, i, y, loss_acc=acc, batch_rows_only=True)
    out = (ctypes.c_ulonglong * 16)()
    assert lib.spex_debug_batch_stamps(out) == 0
    t = np.array(list(out)[:7], np.float64)
    if k >= 10:
        rows.append(np.diff(t) * 0.01)
rows = np.array(rows)
deg = np.diff(csr[0])
print("sample 8: user degree %d, item degree %d" % (deg[int(u[8])], deg[3186 + int(i[8])]))
for n_, m, lo, hi in zip(names, rows.mean(0), rows.min(0), rows.max(0)):
    print("%-32s %.2f us (%.2f-%.2f)" % (n_, m, lo, hi))
print("total stamped %.2f us" % rows.sum(1).mean())
