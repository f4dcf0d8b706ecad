import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
g = SpexGraph(*csr)
rng = np.random.default_rng(0)
E0 = torch.from_numpy(xavier_uniform_np(15593, 64, rng)).cuda()
st = LightGCNStepper(g, E0, 3186)
for _ in range(50): st.propagate()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    s.record()
    for _ in range(200): st.propagate()
    e.record(); e.synchronize()
    best = min(best, s.elapsed_time(e) / 200)
print("SPEX_NT=%s propagate 3 layers: %.2f us" % (os.environ.get("SPEX_NT"), best * 1e3))
