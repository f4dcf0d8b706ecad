"""50 exact training steps (one-call form) + 50 NGCF stepper steps on Epinion2, for rocprofv3 --kernel-trace."""
import argparse, os, sys
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj, ngcf_norm_adj
from spex_amd.ngcf import NGCF
from spex_amd.trainer import LightGCNStepper, NGCFStepper
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
rng = np.random.default_rng(0)
E0 = torch.from_numpy(np.concatenate([xavier_uniform_np(3186, 64, rng), xavier_uniform_np(12407, 64, rng)])).to(dev)
st = LightGCNStepper(SpexGraph(*csr, device=dev), E0, 3186, n_layers=3, lr=1e-3)
u = torch.randint(0, 3185, (256,), device=dev); i = torch.randint(0, 12407, (256,), device=dev)
y = (torch.rand(256, device=dev) < 1 / 6).float()
acc = torch.zeros(1, device=dev)
for _ in range(50):
    st.step_bce(u, i, y, loss_acc=acc, batch_rows_only=True)
nc = ngcf_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
net = NGCF({"n_users": 3185, "n_items": 12407, "norm_adj": sp.csr_matrix((nc[2], nc[1], nc[0]), shape=(15592, 15592))}, dev,
           argparse.Namespace(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")).to(dev)
nst = NGCFStepper(net)
for _ in range(50):
    nst.step(u, i, y, loss_acc=acc)
torch.cuda.synchronize()
# round 3: the north-star step as one call (layer mean left to the BPR kernel) and the deterministic exact step
tu = torch.randint(0, 3185, (2048,), device=dev); tp = torch.randint(0, 12407, (2048,), device=dev); tn = torch.randint(0, 12407, (2048,), device=dev)
for _ in range(50):
    st.step_bpr_sgd(tu, tp, tn)
det = LightGCNStepper(SpexGraph(*csr, device=dev), E0.clone(), 3186, n_layers=3, lr=1e-3, deterministic=True)
for _ in range(50):
    det.step_bce(u, i, y, loss_acc=acc, batch_rows_only=True)
dnst = NGCFStepper(net, deterministic=True)
for _ in range(50):
    dnst.step(u, i, y, loss_acc=acc)
torch.cuda.synchronize()
