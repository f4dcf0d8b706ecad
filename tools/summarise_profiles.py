#!/usr/bin/env python3
"""Turn gpurun_out/prof_r02/ (tools/collect_profiles.sh) into the files profiles/r02/ keeps:
  kernel_stats.csv, kernel_trace_by_grid.csv (bench run), pmc_bpr_raw.json (per kernel: mean counter values and duration)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r02"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "bench", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "bench", "**", "*kernel_trace.csv"), recursive=True):
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "trace_by_grid.py"), f, "spmm", "bpr_", "score_bce",
                          "adam_kernel", "ngcf_layer", "lightgcn_batch", "trust_", "dual_task"], capture_output=True, text=True).stdout
    open(os.path.join(dst, "kernel_trace_by_grid.csv"), "w").write(out)
if os.path.exists(os.path.join(src, "bench_under_rocprof.json")):
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "bench_under_rocprof.json"))
pmc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
        if "bpr" not in name:
            continue
        pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (f, r.get("Dispatch_Id"))
        if key not in seen and r.get("End_Timestamp"):
            seen.add(key)
            dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
summary = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, n_dispatches=max(len(v) for v in cs.values()),
                   dur_ns_under_pmc=(sum(dur[k]) / len(dur[k]) if dur[k] else None)) for k, cs in pmc.items()}
if summary:                                     # (a collection without --pmc passes keeps the stored counters)
    json.dump(summary, open(os.path.join(dst, "pmc_bpr_raw.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
