#!/usr/bin/env python3
"""One Epinion2 epoch through spex_amd.trainer.train_epoch (reference sampler + DataLoader order, steps on the device)."""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spex_amd", "dropin"))
sys.argv = ["x"]
from spex_amd.datasets import materialise_epinion2
from spex_amd.trainer import LightGCNStepper, train_epoch
import lg_parser, utility1.dataloader as dl, utility1.model as model, utility1.utils as utils
from utility1.batch_test import test
root = materialise_epinion2(tempfile.mkdtemp())
args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", root])
utils.set_seed(args.seed)
ds = dl.Loader(args)
net = model.LightGCN(args, ds).cuda()
td = dl.LightTrainData(ds.rec_train_data, ds.m_item, ds.train_mat)
E0 = net.flat_table()                                                                 # the model's own parameters, in place
st = LightGCNStepper(net.Graph, E0, net.num_users + 1, n_layers=net.n_layers, lr=args.lr)
for ep in range(3):
    t0 = time.perf_counter()
    td.ng_sample()
    t1 = time.perf_counter()
    loss = train_epoch(st, td, resample=False).item()
    t2 = time.perf_counter()
    with torch.no_grad():
        net.eval()
        ret = test(net, ds.testRatings, ds.testNegatives)
    t3 = time.perf_counter()
    print("epoch %d: sample %.2f s, train %.2f s (%d steps, %.0f us/step), test %.2f s, loss %.2f HR@10 %.4f"
          % (ep, t1 - t0, t2 - t1, (len(td) + 255) // 256, (t2 - t1) / ((len(td) + 255) // 256) * 1e6, t3 - t2, loss, ret["recall"][0]))

# ---- the same three epochs with the next epoch's sampling and shuffle prepared on a second host thread beside the GPU (train_epochs)
from spex_amd.trainer import train_epochs
torch.cuda.synchronize()
t0 = time.perf_counter()
tot = train_epochs(st, td, 3)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("train_epochs, 3 epochs with overlapped sampling: %.2f s per epoch (sequential: sample + train above); losses %s" % (dt / 3, ["%.2f" % t for t in tot]))
