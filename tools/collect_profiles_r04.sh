#!/bin/bash
# Round-4 profile collection (on the GPU box through gpurun; rocprofv3 gets the program itself after `--`, counters in their own
# passes with --kernel-trace only).  Output under gpurun_out/prof_r04/ (scratch); tools/summarise_profiles_r04.py condenses it into
# profiles/r04/.  Usage: collect_profiles_r04.sh [part ...]   parts: bench steps pmc dual weibo ngcf misc   (default: all)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04
mkdir -p $OUT
PARTS=${*:-bench steps pmc dual weibo ngcf misc}
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  echo "== kernel trace of bench.py" | tee -a $OUT/log.txt
  timeout -k 10 560 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>> $OUT/log.txt || echo "bench trace failed" >> $OUT/log.txt
fi
if has steps; then
  echo "== kernel trace of the steps" | tee -a $OUT/log.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/steps -o steps -- python3 $R/tools/step_prof.py >> $OUT/log.txt 2>&1 || echo "step trace failed" >> $OUT/log.txt
fi
if has pmc; then
  for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    tag=$(echo $pass | tr ' ' '_')
    echo "== pmc epinion2 $pass" | tee -a $OUT/log.txt
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_epinion2/$tag -o spmm -- python3 $R/tools/prof_spmm.py epinion2 20 >> $OUT/log.txt 2>&1 || echo "pmc pass $tag failed" >> $OUT/log.txt
  done
fi
if has dual; then
  echo "== dual-task step: kernel trace + the partitioned one-call step at world 1" | tee -a $OUT/log.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dual_step -o dual -- python3 $R/tools/dual_ab.py --steps 300 --reps 2 >> $OUT/log.txt 2>&1 || echo "dual step trace failed" >> $OUT/log.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dual_part -o dp -- python3 $R/tools/dual_part_time.py > $OUT/dual_part_time.txt 2>> $OUT/log.txt || echo "dual part trace failed" >> $OUT/log.txt
fi
if has weibo; then
  echo "== weibo shape: steps (kernel trace), plain / epilogue launches, edge dropout with the hub fold in the launch vs the fix-up form" | tee -a $OUT/log.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/weibo -o w -- python3 $R/tools/weibo_step_time.py > $OUT/weibo_step_time.txt 2>> $OUT/log.txt || echo "weibo trace failed" >> $OUT/log.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/masked_fold -o mf -- python3 $R/tools/masked_fold_time.py > $OUT/masked_fold_time.txt 2>> $OUT/log.txt || echo "masked fold trace failed" >> $OUT/log.txt
  (cd $R && timeout -k 10 200 python3 tools/plain_spmm_time.py 2>> $OUT/log.txt | grep "us" > $OUT/plain_spmm_time.txt) || echo "plain spmm failed" >> $OUT/log.txt
  (cd $R && timeout -k 10 200 python3 tools/epi_structure_probe.py 2>> $OUT/log.txt | grep "us" > $OUT/epi_structure_probe.txt) || echo "epi probe failed" >> $OUT/log.txt
fi
if has ngcf; then
  echo "== NGCF: dense layer backward (4 waves per tile) + the stepper" | tee -a $OUT/log.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ngcf_bwd -o nb -- python3 $R/tools/ngcf_bwd_dense_time.py > $OUT/ngcf_bwd_dense_time.txt 2>> $OUT/log.txt || echo "ngcf bwd trace failed" >> $OUT/log.txt
fi
if has misc; then
  echo "== bench as the driver runs it (--steps 20 --warmup 5) and the default" | tee -a $OUT/log.txt
  (cd $R && timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2>> $OUT/log.txt) || echo "bench steps20 failed" >> $OUT/log.txt
  (cd $R && timeout -k 10 500 python3 bench.py > $OUT/bench.json 2>> $OUT/log.txt) || echo "bench default failed" >> $OUT/log.txt
fi
echo done | tee -a $OUT/log.txt
