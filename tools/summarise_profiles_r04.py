#!/usr/bin/env python3
"""Turn gpurun_out/prof_r04/ (tools/collect_profiles_r04.sh) into the files profiles/r04/ keeps."""
import csv
import glob
import os
import shutil
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r04"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04"
here = os.path.dirname(os.path.abspath(__file__))
os.makedirs(dst, exist_ok=True)
for sub, name in (("bench", "kernel_stats.csv"), ("steps", "step_kernel_stats.csv"), ("dual_step", "dual_step_kernel_stats.csv"),
                  ("dual_part", "dual_part_kernel_stats.csv"), ("weibo", "weibo_shape_kernel_stats.csv"),
                  ("masked_fold", "masked_fold_kernel_stats.csv"), ("ngcf_bwd", "ngcf_bwd_dense_kernel_stats.csv")):
    for f in glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if not r[0].startswith(("void at::", "__amd_rocclr_fill")) or float(r[4]) >= 0.5][:40]
        csv.writer(open(os.path.join(dst, name), "w")).writerows(keep)
for sub, name in (("bench", "kernel_trace_by_grid.csv"), ("masked_fold", "masked_fold_trace_by_grid.csv")):
    for f in glob.glob(os.path.join(src, sub, "**", "*kernel_trace.csv"), recursive=True):
        out = subprocess.run([sys.executable, os.path.join(here, "trace_by_grid.py"), f, "spmm", "bpr_", "score_bce", "adam_kernel", "ngcf_layer",
                              "lightgcn_batch", "trust_", "dual_task", "reduce_slots", "gated_batch"], capture_output=True, text=True).stdout
        open(os.path.join(dst, name), "w").write(out)
for f in ("bench_under_rocprof.json", "bench_steps20.json", "bench.json", "dual_part_time.txt", "weibo_step_time.txt", "masked_fold_time.txt",
          "plain_spmm_time.txt", "epi_structure_probe.txt", "ngcf_bwd_dense_time.txt"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        txt = "".join(ln for ln in open(p) if "amdgpu.ids" not in ln and not ln.startswith(("W2026", "E2026", "I2026")))
        # (profiles/r04/dual_part_time.txt is the event-timed run WITHOUT the profiler, kept with its history: the profiled run's output
        #  goes beside it)
        open(os.path.join(dst, "dual_part_time_under_rocprof.txt" if f == "dual_part_time.txt" else f), "w").write(txt)
if os.path.isdir(os.path.join(src, "pmc_epinion2")):
    out = subprocess.run([sys.executable, os.path.join(here, "pmc_spmm_summary.py"), os.path.join(src, "pmc_epinion2"), "epinion2_r04"],
                         capture_output=True, text=True).stdout
    open(os.path.join(dst, "pmc_spmm_epinion2_raw.json"), "w").write(out)
    print(out)
if os.path.isdir("gpurun_out/pmc_2e24") and glob.glob("gpurun_out/pmc_2e24/**/*counter_collection.csv", recursive=True):
    out = subprocess.run([sys.executable, os.path.join(here, "pmc_spmm_summary.py"), "gpurun_out/pmc_2e24", "hbm_2e24_r04"],
                         capture_output=True, text=True).stdout
    open(os.path.join(dst, "pmc_spmm_2e24_raw.json"), "w").write(out)
    print(out)
