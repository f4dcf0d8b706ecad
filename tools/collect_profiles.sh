#!/bin/bash
# Round-2 profile collection (run on the GPU box through gpurun; rocprofv3 gets the program itself after `--`):
#   1. kernel trace + stats of the headline bench command
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE + L2 hit/miss | fabric read/write requests) over the BPR step
#      at T = 2^20 (tools/bpr_prof.py: grouped form and atomic form on Epinion2's tables)
# Output under gpurun_out/prof_r02/ (scratch); tools/summarise_profiles.py turns it into what profiles/r02/ keeps.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace of bench.py" | tee $OUT/log.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --hbm-log2-nodes 23 > $OUT/bench_under_rocprof.json 2>> $OUT/log.txt || echo "bench trace failed" >> $OUT/log.txt
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  echo "== pmc $pass" | tee -a $OUT/log.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -o bpr -- python3 $R/tools/bpr_prof.py >> $OUT/log.txt 2>&1 || echo "pmc pass $tag failed" >> $OUT/log.txt
done
find $OUT -name "*.csv" | head -30 | tee -a $OUT/log.txt
echo done | tee -a $OUT/log.txt
