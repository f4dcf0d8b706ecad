"""The one-call dual-task step (spex_dual_task_step_f32) on Epinion2, B = 256, 15 paths: µs per step by HIP events, training-
shaped batches optional.

    python tools/dual_ab.py                           # the default form: two streams, fork / join
    SPEX_DUAL_PIPELINED=1 python tools/dual_ab.py     # the pipelined form of the two-stream step (SPEX_STEP_PIPELINED)
    SPEX_LIB=<other build> python tools/dual_ab.py    # A/B against another build of the library

Under rocprofv3 (`--kernel-trace --stats`) pass --steps 300 to keep the trace small.
"""
import argparse, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spex_amd", "dropin"))
if os.environ.get("SPEX_LIB"):
    from spex_amd import _lib as _l
    _l.LIB_PATH = os.environ["SPEX_LIB"]          # A/B against another build of the library
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--paths", type=int, default=15)
cli = ap.parse_args()
sys.argv = ["x"]
from spex_amd.datasets import materialise_epinion2
import lg_parser, utility1.dataloader as dl, utility1.model_expert_s as mex
from spex_amd.trainer import DualTaskStepper

root = materialise_epinion2(tempfile.mkdtemp())
args = lg_parser.parse_args_r(["--dataset", "epinion2", "--data_path", root])
ds = dl.Loader(args)
net = mex.LightGCN(args, ds).cuda()
dev = torch.device("cuda")
rng = np.random.default_rng(0)
B, T, P_LEN, n_user, m_item = 256, cli.paths, 6, ds.n_users, ds.m_items
u = torch.from_numpy(rng.integers(0, n_user, B)).to(dev); i = torch.from_numpy(rng.integers(0, m_item, B)).to(dev)
y = torch.from_numpy((rng.random(B) < 1 / 6).astype(np.float32)).to(dev)
plen = rng.integers(2, P_LEN + 1, T)
pseq = np.full((T, P_LEN), n_user, dtype=np.int64)
for r, l in enumerate(plen):
    pseq[r, :l] = rng.choice(n_user, size=l, replace=False)
seq, seq_l = torch.from_numpy(pseq).to(dev), torch.from_numpy(plen.astype(np.int64)).to(dev)
tgt = torch.from_numpy(rng.integers(0, n_user, T)).to(dev)
st = DualTaskStepper(net, path_capacity=T, path_len=P_LEN, lr=1e-3, pipelined=os.environ.get("SPEX_DUAL_PIPELINED", "0") == "1")
for _ in range(50):
    st.step(u, i, y, seq, seq_l, tgt)
st.join(); torch.cuda.synchronize()
res = []
for _ in range(cli.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(cli.steps):
        st.step(u, i, y, seq, seq_l, tgt)
    st.join()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / cli.steps * 1e3)
sw = {k: os.environ[k] for k in sorted(os.environ) if k.startswith("SPEX_")}
print("dual-task step, %d paths, %s: us/step min %.1f median %.1f  (%s)  switches %s" % (
    T, ("two streams, pipelined" if st.pipelined else "two streams, fork/join") if st._side is not None else "one stream", min(res), sorted(res)[len(res) // 2],
    " ".join("%.1f" % r for r in res), sw))
