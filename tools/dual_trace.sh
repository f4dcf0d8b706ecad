#!/bin/bash
# Kernel trace of the one-call dual-task step (tools/dual_ab.py) -> gpurun_out/dual_trace/<tag>_kernel_stats.csv ; usage: dual_trace.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-default}
OUT=$R/gpurun_out/dual_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG -o dual -- python3 $R/tools/dual_ab.py --steps 300 --reps 2 > $OUT/$TAG.txt 2>&1 || echo "trace failed" >> $OUT/$TAG.txt
f=$(find $OUT/$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $OUT/${TAG}_kernels.txt <<'P'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    nm = re.sub(r"^void ", "", nm).split("(")[0]
    print("%-60s calls %6s avg %9.1f ns  min %8s max %8s  %5s %%" % (nm[:60], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"][:5]))
P
rm -rf $OUT/$TAG
tail -1 $OUT/$TAG.txt; cat $OUT/${TAG}_kernels.txt
