#!/usr/bin/env python3
"""Turn gpurun_out/prof_r03/ (tools/collect_profiles_r03.sh) into the files profiles/r03/ keeps."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r03"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03"
here = os.path.dirname(os.path.abspath(__file__))
os.makedirs(dst, exist_ok=True)
for sub, name in (("bench", "kernel_stats.csv"), ("steps", "step_kernel_stats.csv"), ("dual", "dual_task_kernel_stats.csv")):
    for f in glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, name))
for f in glob.glob(os.path.join(src, "bench", "**", "*kernel_trace.csv"), recursive=True):
    out = subprocess.run([sys.executable, os.path.join(here, "trace_by_grid.py"), f, "spmm", "bpr_", "score_bce", "adam_kernel", "ngcf_layer",
                          "lightgcn_batch", "trust_", "dual_task", "reduce_slots"], capture_output=True, text=True).stdout
    open(os.path.join(dst, "kernel_trace_by_grid.csv"), "w").write(out)
for f in glob.glob(os.path.join(src, "dual_step", "**", "*kernel_stats.csv"), recursive=True):      # the one-call dual-task step alone
    shutil.copy(f, os.path.join(dst, "dual_step_kernel_stats.csv"))
for f in ("bench_under_rocprof.json", "dual_task_time.txt", "dual_step_forms.txt", "trust_forms.txt", "trust_stamps.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
lines = []
for d in sorted(glob.glob(os.path.join(src, "trust_alone", "*"))):                  # the trust head alone: its launches' durations
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "trust_" in r["Name"]:
                lines.append("%-10s %-28s calls %4s  avg %7.2f us  min %7.2f  max %7.2f" % (
                    os.path.basename(d), r["Name"].split("::")[-1].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3,
                    float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
if lines:
    open(os.path.join(dst, "trust_kernels_alone.txt"), "w").write(
        "spex_trust_head_train_f32 alone (tools/trust_forms_time.py one <users> <paths> under rocprofv3 --kernel-trace --stats):\n" + "\n".join(lines) + "\n")
if os.path.isdir(os.path.join(src, "pmc_epinion2")):
    out = subprocess.run([sys.executable, os.path.join(here, "pmc_spmm_summary.py"), os.path.join(src, "pmc_epinion2"), "epinion2_r03"],
                         capture_output=True, text=True).stdout
    open(os.path.join(dst, "pmc_spmm_epinion2_raw.json"), "w").write(out)
    print(out)
# push balance: the k-th lightgcn_batch_kernel<true> dispatch of the probe's trace = the k-th row of its features file
feat = os.path.join(src, "push_features.csv")
traces = glob.glob(os.path.join(src, "push", "**", "*kernel_trace.csv"), recursive=True)
if os.path.exists(feat) and traces:
    rows = [r for r in csv.DictReader(open(traces[0])) if "lightgcn_batch_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = np.asarray([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
    F = np.loadtxt(feat, delimiter=",", skiprows=1)
    n = min(len(dur), len(F))
    dur, F = dur[20:n], F[20:n]                     # (warm-up launches dropped)
    A = np.stack([np.ones(len(dur)), F[:, 1]], 1)
    c1, *_ = np.linalg.lstsq(A, dur, rcond=None)
    A2 = np.stack([np.ones(len(dur)), F[:, 1], F[:, 3]], 1)
    c2, *_ = np.linalg.lstsq(A2, dur, rcond=None)
    ideal = F[:, 3] / (256 * 3 * 16.0)              # every wave of the launch issuing an equal share of the batch's runs
    txt = ("lightgcn_batch_kernel on %d training-shaped Epinion2 batches (B = 256; rocprofv3 kernel-trace durations joined with the host's\n"
           "per-batch run counts, tools/push_balance_probe.py):\n"
           "  launch us: mean %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f\n"
           "  runs of 16 entries per batch: mean %.0f; longest sample: mean %.0f runs; largest number of runs one wave issues: mean %.2f, max %.0f\n"
           "  fit  us = %.2f + %.2f * max_runs_per_wave                       (correlation %.2f)\n"
           "  fit  us = %.2f + %.2f * max_runs_per_wave + %.4f * runs_in_batch\n"
           "  a perfectly balanced push (every wave issuing runs_in_batch / 12 288) would issue %.2f runs per wave on average instead of\n"
           "  the measured maximum %.2f: by the first fit that is worth %.2f us of the %.2f us launch\n"
           % (len(dur), dur.mean(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), F[:, 3].mean(), F[:, 4].mean(),
              F[:, 1].mean(), F[:, 1].max(), c1[0], c1[1], np.corrcoef(dur, F[:, 1])[0, 1], c2[0], c2[1], c2[2], ideal.mean(), F[:, 1].mean(),
              c1[1] * (F[:, 1].mean() - ideal.mean()), dur.mean()))
    open(os.path.join(dst, "push_balance.txt"), "w").write(txt)
    print(txt)
