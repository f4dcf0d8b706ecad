#!/usr/bin/env python3
"""NGCF (one layer, the reference's default) epochs on Epinion2 through spex_amd.trainer: the sampler (Data.sample_epoch: blocked replay
of the reference's `random` stream; fast=False: draw by draw), one epoch as one native call (train_epoch_ngcf), and three epochs with
the next epoch's samples prepared beside the GPU (train_epochs_ngcf).  Wall-clock seconds."""
import argparse, os, random, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["x"]
from spex_amd.dropin.ngcf.utility.load_data import Data
from spex_amd.ngcf import NGCF
from spex_amd.trainer import NGCFStepper, train_epoch_ngcf, train_epochs_ngcf
e = np.load(os.path.join(ROOT, "tests", "golden", "epinion2_dataset.npz"))
root = tempfile.mkdtemp()
rec = os.path.join(root, "epinion2", "rec")
os.makedirs(rec)
pairs = e["train"].astype(np.int64)
pairs = pairs[np.argsort(pairs[:, 0], kind="stable")]
with open(os.path.join(rec, "train.txt"), "w") as f:
    users, start = np.unique(pairs[:, 0], return_index=True)
    for k, u in enumerate(users):
        end = start[k + 1] if k + 1 < len(users) else len(pairs)
        f.write(str(u) + "".join(" %d" % i for i in pairs[start[k]:end, 1]) + "\n")
with open(os.path.join(rec, "test.txt"), "w") as f:
    for u, p in zip(e["test_users"].astype(int), e["test_pos"].astype(int)):
        f.write("%d %d\n" % (u, p))
with open(os.path.join(rec, "negative.txt"), "w") as f:
    for u, negs in zip(e["test_users"].astype(int), e["test_neg"].astype(np.int64)):
        f.write(str(u) + "".join(" %d" % i for i in negs) + "\n")
random.seed(2020); torch.manual_seed(2020)
data = Data(path=os.path.join(root, "epinion2"), batch_size=256)
_, norm, _ = data.get_adj_mat()
args = argparse.Namespace(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")
model = NGCF({"n_users": data.n_users, "n_items": data.n_items, "norm_adj": norm}, "cuda", args).cuda()
model.train()
st = NGCFStepper(model, lr=1e-3)
for fast in (False, True):
    t0 = time.perf_counter(); u, v, r = data.sample_epoch(fast=fast); t1 = time.perf_counter()
    print("sample_epoch(fast=%s): %.2f s for %d samples" % (fast, t1 - t0, len(u)), flush=True)
train_epoch_ngcf(st, data).item()                                   # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
tot = train_epoch_ngcf(st, data).item()
t1 = time.perf_counter()
print("train_epoch_ngcf (sampling + one native call): %.2f s, loss %.2f" % (t1 - t0, tot), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
tots = train_epochs_ngcf(st, data, 3)
t1 = time.perf_counter()
print("train_epochs_ngcf, 3 epochs with overlapped sampling: %.2f s per epoch (%d steps each)" % ((t1 - t0) / 3, st.t // 5))
