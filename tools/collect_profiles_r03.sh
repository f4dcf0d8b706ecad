#!/bin/bash
# Round-3 profile collection (on the GPU box through gpurun; rocprofv3 gets the program itself after `--`, counters in their own
# passes).  Output under gpurun_out/prof_r03/ (scratch); tools/summarise_profiles_r03.py condenses it into profiles/r03/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace of bench.py" | tee $OUT/log.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>> $OUT/log.txt || echo "bench trace failed" >> $OUT/log.txt
echo "== kernel trace of the steps" | tee -a $OUT/log.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/steps -o steps -- python3 $R/tools/step_prof.py >> $OUT/log.txt 2>&1 || echo "step trace failed" >> $OUT/log.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dual -o dual -- python3 $R/tools/dual_task_time.py > $OUT/dual_task_time.txt 2>> $OUT/log.txt || echo "dual trace failed" >> $OUT/log.txt
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  echo "== pmc epinion2 $pass" | tee -a $OUT/log.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_epinion2/$tag -o spmm -- python3 $R/tools/prof_spmm.py epinion2 20 >> $OUT/log.txt 2>&1 || echo "pmc pass $tag failed" >> $OUT/log.txt
done
echo "== push balance" | tee -a $OUT/log.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/push -o push -- python3 $R/tools/push_balance_probe.py $OUT/push_features.csv >> $OUT/log.txt 2>&1 || echo "push probe failed" >> $OUT/log.txt
echo "== dual-task step forms, trust head forms, trust kernel phase stamps" | tee -a $OUT/log.txt
cd $R
(timeout -k 10 200 python3 tools/dual_ab.py && SPEX_DUAL_PIPELINED=1 timeout -k 10 200 python3 tools/dual_ab.py && SPEX_DUAL_ONE_STREAM=1 timeout -k 10 200 python3 tools/dual_ab.py \
  && SPEX_DUAL_FUSED_MIDDLE=0 timeout -k 10 200 python3 tools/dual_ab.py) 2>&1 | grep "dual-task step" > $OUT/dual_step_forms.txt || echo "dual_ab failed" >> $OUT/log.txt
timeout -k 10 600 python3 tools/trust_forms_time.py > $OUT/trust_forms.txt 2>> $OUT/log.txt || echo "trust forms failed" >> $OUT/log.txt
if [ -f spex_amd/lib/libspexhip_stamps.so ]; then
  timeout -k 10 200 python3 tools/trust_stamps.py 2>> $OUT/log.txt | grep -v amdgpu.ids > $OUT/trust_stamps.txt || echo "trust stamps failed" >> $OUT/log.txt
fi
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dual_step -o dual -- python3 $R/tools/dual_ab.py --steps 300 --reps 2 >> $OUT/log.txt 2>&1 || echo "dual step trace failed" >> $OUT/log.txt
for sz in "3185 15" "6812 15"; do      # the trust head's two launches on their own (the library's choice of form)
  n=$(echo $sz | tr ' ' 'x')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trust_alone/$n -o t -- python3 $R/tools/trust_forms_time.py one $sz >> $OUT/log.txt 2>&1 || echo "trust alone $n failed" >> $OUT/log.txt
done
echo done | tee -a $OUT/log.txt
