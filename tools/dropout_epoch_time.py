#!/usr/bin/env python3
"""The reference's recommended LightGCN configuration (`--dropout 1 --keepprob 0.3`, README.md:119-123) on Epinion2: us per
training step (a fresh edge mask on both handles + spex_lightgcn_step_bce_f32, as trainer.train_epoch issues them) with the mask
off / drawn in-kernel (Philox) / replayed from the reference's CPU stream (torch.rand(nnz) per step), and the launch-by-launch
form of the masked step for comparison."""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spex_amd", "dropin"))
sys.argv = ["x"] + sys.argv[1:]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sys.argv = ["x"]
from spex_amd.datasets import materialise_epinion2
from spex_amd.trainer import LightGCNStepper, edge_dropout_mask
import lg_parser, utility1.dataloader as dl, utility1.model as model, utility1.utils as utils
root = materialise_epinion2(tempfile.mkdtemp())
args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", root])
utils.set_seed(args.seed)
ds = dl.Loader(args)
net = model.LightGCN(args, ds).cuda()
dev = torch.device("cuda")
rng = np.random.default_rng(0)
B = 256
u = torch.from_numpy(rng.integers(0, ds.n_users, (64, B))).to(dev); i = torch.from_numpy(rng.integers(0, ds.m_items, (64, B))).to(dev)
y = torch.from_numpy((rng.random((64, B)) < 1 / 6).astype(np.float32)).to(dev)
E0 = net.flat_table().detach()
acc = torch.zeros(1, device=dev)
for name, stream, one_call in (("no dropout, one-call step", None, True), ("Philox mask, one-call step", "philox", True),
                               ("Philox mask, launch-by-launch step", "philox", False), ("reference CPU stream, one-call step", "reference", True)):
    st = LightGCNStepper(net.Graph, E0.clone(), net.num_users + 1, n_layers=net.n_layers, lr=args.lr,
                         graph_t=net.Graph if stream is None else net._transposed())
    def step(k):
        if stream is not None:
            mask = edge_dropout_mask(st.graph, 0.3, stream, 7, k + 1)
            st.graph.set_edge_mask(*mask)
            st.graph_t.set_edge_mask(*mask)
        st.step_bce(u[k & 63], i[k & 63], y[k & 63], loss_acc=acc, batch_rows_only=one_call)
    for k in range(50): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    torch.cuda.synchronize()
    print("%-44s %.1f us/step" % (name, (time.perf_counter() - t0) / steps * 1e6))
    st.graph.set_edge_mask(0); st.graph_t.set_edge_mask(0)

# ---- the dual-task step (main_auto_expert_s.py) under the same dropout: the rec branch's handles masked, 15 paths per step
import utility1.model_expert_s as mex
from spex_amd.trainer import DualTaskStepper
for name, drop in (("dual-task step, no dropout", False), ("dual-task step, Philox mask", True)):
    dargs = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", root] + (["--dropout", "1", "--keepprob", "0.3"] if drop else []))
    utils.set_seed(dargs.seed)
    dnet = mex.LightGCN(dargs, ds).cuda()
    T, P_LEN = 15, 6
    dst = DualTaskStepper(dnet, path_capacity=T, path_len=P_LEN, lr=1e-3)
    plen = rng.integers(2, P_LEN + 1, T)
    seq = np.full((T, P_LEN), ds.n_users, dtype=np.int64)
    for r, l in enumerate(plen):
        seq[r, :l] = rng.choice(ds.n_users, size=l, replace=False)
    seq_d, len_d = torch.from_numpy(seq).to(dev), torch.from_numpy(plen.astype(np.int64)).to(dev)
    tgt = torch.from_numpy(rng.integers(0, ds.n_users, T)).to(dev)
    def dstep(k):
        if drop:
            dst.set_edge_dropout(edge_dropout_mask(dnet.Graph, 0.3, "philox", 7, k + 1))
        dst.step(u[k & 63], i[k & 63], y[k & 63], seq_d, len_d, tgt)
    for k in range(50): dstep(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps): dstep(k)
    torch.cuda.synchronize()
    print("%-44s %.1f us/step" % (name, (time.perf_counter() - t0) / steps * 1e6))
    dst.set_edge_dropout(None)
