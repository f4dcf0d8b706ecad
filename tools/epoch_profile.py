#!/usr/bin/env python3
"""Where does an epoch of the reference-shaped driver (tests/drivers/rec_driver.py) spend its host time?  Runs one
epoch of Epinion2 through the drop-in modules under cProfile and prints phase timings + the top host functions."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spex_amd", "dropin"))
from spex_amd.datasets import materialise_epinion2   # noqa: E402

root = materialise_epinion2(tempfile.mkdtemp())
sys.argv = ["rec_driver.py", "--dataset", "epinion2", "--data_path", root, "--epochs", "1"]
sys.path.insert(0, os.path.join(ROOT, "tests", "drivers"))
import rec_driver as D   # noqa: E402
import torch             # noqa: E402

t0 = time.perf_counter()
D.train_loader.dataset.ng_sample()
t1 = time.perf_counter()
D.run_epoch(0)           # warm (includes a second ng_sample)
torch.cuda.synchronize()
t2 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
D.run_epoch(1)
torch.cuda.synchronize()
pr.disable()
t3 = time.perf_counter()
D.evaluate(1)
torch.cuda.synchronize()
t4 = time.perf_counter()
print("ng_sample %.2f s | first epoch %.2f s | profiled epoch %.2f s (%.0f us/step) | test() %.2f s"
      % (t1 - t0, t2 - t1, t3 - t2, (t3 - t2) / len(D.train_loader) * 1e6, t4 - t3))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
