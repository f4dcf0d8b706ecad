import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.sparse as sp
from spex_amd.datasets import load_epinion2
from spex_amd.graph import ngcf_norm_adj
from spex_amd.ngcf import NGCF
from spex_amd.trainer import NGCFStepper
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
n_u, n_i = 3185, 12407
csr = ngcf_norm_adj(tr[:, 0], tr[:, 1], n_u, n_i)
rng = np.random.default_rng(2)
ub = torch.from_numpy(rng.integers(0, n_u, 256)).to(dev); ib = torch.from_numpy(rng.integers(0, n_i, 256)).to(dev)
yb = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32)).to(dev)
def timed(fn, n=500):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for layers in ("[64]", "[64,64]"):
    L = layers.count("64")
    for p in (0.1, 0.0):
        a = argparse.Namespace(embed_size=64, layer_size=layers, mess_dropout=str([p] * L), regs="[1e-5]")
        net = NGCF({"n_users": n_u, "n_items": n_i, "norm_adj": sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(n_u + n_i,) * 2)}, dev, a).to(dev)
        net.train()
        st = NGCFStepper(net, lr=1e-3)
        acc = torch.zeros(1, device=dev)
        print("NGCF %d layers, message dropout %.1f: %.1f us per step" % (L, p, timed(lambda: st.step(ub, ib, yb, loss_acc=acc))), flush=True)
