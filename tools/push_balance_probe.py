#!/usr/bin/env python3
"""How much of lightgcn_batch_kernel's launch time is the push's imbalance?  (measure before building a queue.)

For 300 training-shaped batches of Epinion2 (a random observed pair or one of its five same-user negatives: users and positive
items arrive in proportion to their degree) the host computes, per batch, what the kernel's dealing implies — the batch's total
16-entry runs, and the LARGEST number of runs any one wave has to issue (a sample's runs are dealt over up to three 16-wave parts
when it has more than 16 runs per part) — and one hipEvent pair measures that batch's launch.  A least-squares fit
time = a + b * max_runs_per_wave then says what a perfect balance could save (b * (max - mean) per launch)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spex_amd import ops
from spex_amd.datasets import load_epinion2, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
deg = np.diff(csr[0])
g = SpexGraph(*csr, device=dev)
n, n_u, L, B = len(deg), 3186, 3, 256
rng = np.random.default_rng(0)
X = torch.from_numpy(np.concatenate([xavier_uniform_np(3186, 64, rng), xavier_uniform_np(12407, 64, rng)])).to(dev)
run = X.clone()
parts, runs_per_part = int(os.environ.get("SPEX_BATCH_PARTS", "3")), int(os.environ.get("SPEX_BATCH_RUNS_PER_PART", "16"))
# Launch durations come from rocprofv3's kernel trace of THIS process (run it as `rocprofv3 --kernel-trace -- python3 tools/
# push_balance_probe.py <features.csv>`): the k-th lightgcn_batch_kernel dispatch is the k-th row of the features file, which
# tools/summarise_profiles_r03.py joins and fits.  (A hipEvent pair around one ~15 us launch issued from Python measures the host.)
feat = open(sys.argv[1] if len(sys.argv) > 1 else "push_features.csv", "w")
feat.write("launch,max_runs_per_wave,mean_runs_per_wave,runs_in_batch,longest_sample_runs\n")
g_out, G = torch.zeros(n, 64, device=dev), torch.zeros(n, 64, device=dev)
loss = torch.zeros(1, device=dev)
for k in range(320):
    idx = rng.integers(0, len(tr), B)
    u, i = tr[idx, 0].copy(), tr[idx, 1].copy()
    neg = rng.random(B) < 5 / 6
    i[neg] = rng.integers(0, 12407, int(neg.sum()))
    y = (~neg).astype(np.float32)
    n_runs = (deg[u] + 15) // 16 + (deg[i + n_u] + 15) // 16
    act = np.clip((n_runs + runs_per_part - 1) // runs_per_part, 1, parts)
    per_wave = np.ceil(n_runs / (act * 16.0))
    ud, idv, yd = (torch.from_numpy(a).to(dev) for a in (u, i, y))
    g_out.zero_(); G.zero_()
    ops.lightgcn_batch(g, X, run, float(L + 1), ud, idv, yd, n_u, 1.0 / B, 1.0 / (L + 1), loss, g_out, G)
    feat.write("%d,%.0f,%.3f,%.0f,%.0f\n" % (k, per_wave.max(), per_wave.mean(), n_runs.sum(), n_runs.max()))
torch.cuda.synchronize()
feat.close()
