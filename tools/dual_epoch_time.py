#!/usr/bin/env python3
"""One dual-task epoch on Epinion2 (main_auto_expert_s.py:53-91) through trainer.train_epoch_dual: where the wall time goes —
sampling, path selection (dual_task_epoch_paths: the reference's random.sample cuts), staging, the 4 906 one-call steps."""
import os, sys, tempfile, time
from collections import defaultdict
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spex_amd", "dropin"))
sys.argv = ["x"]
from spex_amd.datasets import materialise_epinion2
from spex_amd.trainer import DualTaskStepper, dual_task_epoch_paths, dataloader_epoch_order, train_epoch_dual
import lg_parser, utility1.dataloader as dl, utility1.model_expert_s as mex, utility1.utils as utils
from utility2.utils import Data
t = np.load(os.path.join(ROOT, "tests", "golden", "trust_epinion2_paths.npz"))
raw_train = ([r[:l].tolist() for r, l in zip(t["train_paths"].astype(np.int64), t["train_len"])], t["train_targets"].astype(np.int64).tolist())
root = materialise_epinion2(tempfile.mkdtemp())
args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", root])
utils.set_seed(args.seed)
ds = dl.Loader(args)
net = mex.LightGCN(args, ds).cuda()
td = dl.LightTrainData(ds.rec_train_data, ds.m_item, ds.train_mat)
by_user = defaultdict(list)
for k, p in enumerate(raw_train[0]):
    by_user[p[0]].append(k)
train2 = Data(raw_train, ds.n_users, shuffle=False)
cap = 3 * (len(raw_train[0]) // ((len(td) + 255) // 256))
st = DualTaskStepper(net, path_capacity=cap, path_len=train2.len_max, lr=args.lr)
t0 = time.perf_counter(); td.ng_sample(); t1 = time.perf_counter()
order = dataloader_epoch_order(len(td)).numpy(); users_h = td.users_fill[order]
t2 = time.perf_counter()
chosen = dual_task_epoch_paths([users_h[s:s + 256] for s in range(0, len(td), 256)], by_user, cap)
t3 = time.perf_counter()
flat = np.fromiter((k for c in chosen for k in c), dtype=np.int64, count=sum(len(c) for c in chosen))
inputs, mask, targets = train2.get_slice(flat)
t4 = time.perf_counter()
print("ng_sample %.2f s, shuffle %.2f s, path selection %.2f s (%d paths), get_slice %.2f s" % (t1 - t0, t2 - t1, t3 - t2, len(flat), t4 - t3), flush=True)
for ep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tot = train_epoch_dual(st, td, train2, by_user, cap).cpu().numpy()
    t1 = time.perf_counter()
    print("train_epoch_dual: %.2f s (%d steps, %.0f us per step all in), losses %s" % (t1 - t0, st.t // (ep + 1), (t1 - t0) / (st.t / (ep + 1)) * 1e6, tot), flush=True)
from spex_amd.trainer import train_epochs_dual
torch.cuda.synchronize(); t0 = time.perf_counter()
tots = train_epochs_dual(st, td, train2, by_user, cap, 3)
t1 = time.perf_counter()
print("train_epochs_dual, 3 epochs with the next epoch prepared beside the native call: %.2f s per epoch" % ((t1 - t0) / 3))
