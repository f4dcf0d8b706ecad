"""Host vs GPU time of the exact training step's two forms on Epinion2 (diagnostic)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
rng = np.random.default_rng(0)
E0 = torch.from_numpy(np.concatenate([xavier_uniform_np(3186, 64, rng), xavier_uniform_np(12407, 64, rng)])).to(dev)
g = SpexGraph(*csr, device=dev)
st = LightGCNStepper(g, E0, 3186, n_layers=3, lr=1e-3)
u = torch.randint(0, 3185, (256,), device=dev); i = torch.randint(0, 12407, (256,), device=dev)
y = (torch.rand(256, device=dev) < 1 / 6).float()
acc = torch.zeros(1, device=dev)
import gc; gc.collect(); gc.freeze()
for name, fn in (("full", lambda: st.step_bce(u, i, y, loss_acc=acc)), ("rows_only", lambda: st.step_bce(u, i, y, loss_acc=acc, batch_rows_only=True)),
                 ("propagate", st.propagate), ("bpr2048", lambda: st.step_bpr_sgd(u, i, i)), ("ngcf_stepper", None)):
    if name == "ngcf_stepper":
        import argparse, scipy.sparse as sp
        from spex_amd.graph import ngcf_norm_adj
        from spex_amd.ngcf import NGCF
        from spex_amd.trainer import NGCFStepper
        nc = ngcf_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
        net = NGCF({"n_users": 3185, "n_items": 12407, "norm_adj": sp.csr_matrix((nc[2], nc[1], nc[0]), shape=(15592, 15592))}, dev,
                   argparse.Namespace(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")).to(dev)
        nst = NGCFStepper(net)
        fn = lambda: nst.step(u, i, y, loss_acc=acc)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(500): fn()
    e.record(); t_host = time.perf_counter() - t0
    e.synchronize()
    print(name, "gpu us/step %.1f  host-issue us/step %.1f" % (s.elapsed_time(e) * 2, t_host / 500 * 1e6), flush=True)
