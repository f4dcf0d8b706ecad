import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
from spex_amd.datasets import xavier_uniform_np
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
nu, ni = 40, 60
u = rng.integers(0, nu, 300); i = rng.integers(0, ni, 300)
csr = lightgcn_norm_adj(u, i, nu, ni)
g = SpexGraph(*csr, device=dev)
E0 = torch.from_numpy(np.concatenate([xavier_uniform_np(nu + 1, 64, rng), xavier_uniform_np(ni, 64, rng)])).to(dev)
st = LightGCNStepper(g, E0, nu + 1, n_layers=3, lr=1e-3)
tu = torch.from_numpy(rng.integers(0, nu, 64)).to(dev); tp = torch.from_numpy(rng.integers(0, ni, 64)).to(dev); tn = torch.from_numpy(rng.integers(0, ni, 64)).to(dev)
y = torch.zeros(64, device=dev); acc = torch.zeros(1, device=dev)
for name, fn in (("step_bpr_sgd", lambda: st.step_bpr_sgd(tu, tp, tn)), ("step_bce", lambda: st.step_bce(tu, tp, y, loss_acc=acc, batch_rows_only=True))):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(2000): fn()
        dt_issue = time.perf_counter() - t
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("%s on a 100-node graph: host issue %.1f us/call, with drain %.1f us/call" % (name, dt_issue / 2000 * 1e6, dt / 2000 * 1e6))
