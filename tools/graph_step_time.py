"""Is the exact training step host-bound?  Eager vs captured-in-a-HIP-graph replay of the same launch sequence."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
dev = torch.device("cuda:0")
g = SpexGraph(*csr, device=dev)
E0 = torch.from_numpy(xavier_uniform_np(15593, 64, np.random.default_rng(0))).to(dev)
st = LightGCNStepper(g, E0, 3186)
u = torch.randint(0, 3185, (256,), device=dev); i = torch.randint(0, 12407, (256,), device=dev)
y = (torch.rand(256, device=dev) < 1 / 6).float()
tu = torch.randint(0, 3185, (2048,), device=dev); tp = torch.randint(0, 12407, (2048,), device=dev); tn = torch.randint(0, 12407, (2048,), device=dev)
def timeit(fn, it=300):
    for _ in range(20): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
for name, fn in (("exact BCE+Adam step", lambda: st.step_bce(u, i, y)), ("propagate + fused BPR step", lambda: st.step_bpr_sgd(tu, tp, tn))):
    eager = timeit(fn)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    rep = timeit(graph.replay)
    print("%-28s eager %.1f us   graph replay %.1f us" % (name, eager, rep))
