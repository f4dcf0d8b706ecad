"""Where does a dual-task (main_auto_expert_s.py) training step spend its time?  Epinion2 graph, synthetic trust paths."""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spex_amd", "dropin"))
sys.argv = ["x"]
from spex_amd.datasets import materialise_epinion2
import lg_parser, utility1.dataloader as dl, utility1.model_expert_s as mex
from utility2.utils import Data
root = materialise_epinion2(tempfile.mkdtemp())
args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", root])
ds = dl.Loader(args)
net = mex.LightGCN(args, ds).cuda()
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
rng = np.random.default_rng(0)
paths = [rng.choice(3185, size=int(rng.integers(2, 7)), replace=False).tolist() for _ in range(20000)]
targets = rng.integers(0, 3185, 20000).tolist()
trust = Data((paths, targets), ds.n_users)
u = torch.from_numpy(rng.integers(0, 3185, 256)); i = torch.from_numpy(rng.integers(0, 12407, 256))
y = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.int64))
sl = rng.integers(0, 20000, 15)      # the reference driver caps a step at 3 x trust_batch_size = 15 paths
def step(trust_on):
    opt.zero_grad()
    if trust_on:
        l1, l2 = net(u, i, y, sl, trust, flag=0)
        loss = l1 + l2
    else:
        loss = net(u, i, y, None, None, flag=0)[0] if False else torch.nn.functional.binary_cross_entropy_with_logits(
            (net._gated_tables()[0][u.cuda()] * net._gated_tables()[1][i.cuda()]).sum(1), y.cuda().float())
    loss.backward(); opt.step()
net.train()
for name, on in (("rec branch only (gate, autograd, torch Adam)", False), ("rec + trust head", True)):
    for _ in range(10): step(on)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): step(on)
    torch.cuda.synchronize()
    print("%-48s %.2f ms/step" % (name, (time.perf_counter() - t) / 50 * 1e3))

# the same step as one library call (spex_dual_task_step_f32, trainer.DualTaskStepper): 15 paths per step like the driver's cap
from spex_amd.trainer import DualTaskStepper
cap = 15
st = DualTaskStepper(net, path_capacity=cap, path_len=trust.len_max, lr=1e-3)
inputs, mask, tg = trust.get_slice(sl[:cap])
seq = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.int64)).cuda()
seq_l = torch.from_numpy(mask.sum(1).astype(np.int64)).cuda()
tgt = torch.from_numpy(np.asarray(tg).astype(np.int64)).cuda()
uc, ic, yc = u.cuda(), i.cuda(), y.cuda().float()
for _ in range(50): st.step(uc, ic, yc, seq, seq_l, tgt)
torch.cuda.synchronize(); t = time.perf_counter()
n = 500
for _ in range(n): st.step(uc, ic, yc, seq, seq_l, tgt)
torch.cuda.synchronize()
print("%-48s %.3f ms/step" % ("one-call step (DualTaskStepper), 15 paths, %s" % ("two streams" if st._side is not None else "one stream"),
                              (time.perf_counter() - t) / n * 1e3))
