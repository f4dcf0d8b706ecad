#!/bin/bash
# The HBM-resident graph's (Epinion2 x 1076, ~2^24 nodes) counter passes on the current build: separate --pmc passes with
# --kernel-trace only, the program itself after `--`.  Output: gpurun_out/pmc_2e24/<pass>/ ; condense with
#   python tools/pmc_spmm_summary.py gpurun_out/pmc_2e24 hbm_2e24
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_2e24
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  echo "== pmc 2^24 $pass" | tee -a $OUT/log.txt
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$tag -o spmm -- python3 $R/tools/prof_spmm.py hbm 4 24 >> $OUT/log.txt 2>&1 || { echo "pmc pass $tag failed" | tee -a $OUT/log.txt; exit 1; }
done
echo done | tee -a $OUT/log.txt
