"""NGCF one-call step on Epinion2, B = 256, message dropout 0.1: us per step (SPEX_NGCF_TWO_STREAMS=1 for the two-stream form)."""
import argparse, os, sys, time
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spex_amd.datasets import load_epinion2
from spex_amd.graph import ngcf_norm_adj
from spex_amd.ngcf import NGCF
from spex_amd.trainer import NGCFStepper
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
u = torch.randint(0, 3185, (256,), device=dev); i = torch.randint(0, 12407, (256,), device=dev)
y = (torch.rand(256, device=dev) < 1 / 6).float()
acc = torch.zeros(1, device=dev)
nc = ngcf_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
net = NGCF({"n_users": 3185, "n_items": 12407, "norm_adj": sp.csr_matrix((nc[2], nc[1], nc[0]), shape=(15592, 15592))}, dev,
           argparse.Namespace(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")).to(dev)
nst = NGCFStepper(net)
for _ in range(100):
    nst.step(u, i, y, loss_acc=acc)
torch.cuda.synchronize()
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    t0 = time.perf_counter()
    for _ in range(500):
        nst.step(u, i, y, loss_acc=acc)
    host = (time.perf_counter() - t0) / 500 * 1e6
    b.record(); torch.cuda.synchronize()
    print("ngcf step %s: %.2f us (host enqueue %.1f us per step)" % ("two streams" if nst._side is not None else "one stream", a.elapsed_time(b) / 500 * 1e3, host))
