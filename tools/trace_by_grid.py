#!/usr/bin/env python3
"""Group a rocprofv3 kernel-trace CSV by kernel instantiation and grid size (durations in ns):
    python tools/trace_by_grid.py <..._kernel_trace.csv> [name-substring ...] > kernel_trace_by_grid.csv"""
import csv
import statistics
import sys
from collections import defaultdict

pats = sys.argv[2:] or ["spmm", "bpr_kernel", "score_bce", "adam_kernel", "sddmm", "edge_softmax", "ngcf_layer"]
groups = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if not any(p in name for p in pats):
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        key = (short, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]), r.get("VGPR_Count", ""), r.get("SGPR_Count", ""),
               r.get("LDS_Block_Size", ""))
        groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "count", "mean", "median", "min", "max"])
for k in sorted(groups):
    d = groups[k]
    w.writerow(list(k) + [len(d), statistics.mean(d), statistics.median(d), min(d), max(d)])
