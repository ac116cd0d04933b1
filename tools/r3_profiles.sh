#!/bin/bash
# Round-3 profile set (run from the repo root on the GPU box): the round's standard five rocprofv3 processes (tools/profile_round.sh),
# the L2 <-> L1 request counters of the HAB tail before (HAT_TAIL_V2=1: hat_hab_tail) and after (default: hat_hab_tail3), the
# kernel statistics of HAT-L x4 512x512 (BASELINE config 3), and the tail-to-tail gap trace.
set -o pipefail
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
tools/profile_round.sh r03_final > $O/r03_profile_round.log 2>&1 || { tail -5 $O/r03_profile_round.log; exit 1; }
echo "standard set done"
cd /tmp
for arm in ${SKIP_L2:+} $( [ -z "$SKIP_L2" ] && echo v3 v2 ); do
  rm -rf $O/pmcL_$arm
  if [ $arm = v2 ]; then export HAT_TAIL_V2=1; else unset HAT_TAIL_V2; fi
  ( cd $R && timeout -k 10 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmcL_$arm -- python3 tools/run_forward.py > $O/pmcL_$arm.log 2>&1 ) || { tail -5 $O/pmcL_$arm.log; exit 1; }
  ( cd $R && python tools/pmc_summary.py $O/pmcL_$arm > $O/r03_final/r03_final_pmc_l2_$arm.txt )
  rm -rf $O/pmcL_$arm
done
unset HAT_TAIL_V2
echo "L2 passes done"
rm -rf $O/prof_hatl
( cd $R && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hatl -- python3 bench.py --model HAT-L --height 512 --width 512 --steps 5 --warmup 2 --cpu-crop 0 --no-f32-path > $O/r03_final/r03_hatl_bench_under_rocprof.json 2> $O/prof_hatl.err ) || { tail -5 $O/prof_hatl.err; exit 1; }
cp $(ls $O/prof_hatl/*/*kernel_stats.csv | head -1) $O/r03_final/r03_hatl_kernel_stats.csv
find $O/prof_hatl -name "*kernel_trace.csv" -delete
echo "HAT-L stats done"
cd $R
tools/gap_trace.sh > $O/r03_final/r03_gap_trace.txt 2>&1 || { tail -5 $O/r03_final/r03_gap_trace.txt; exit 1; }
cat $O/r03_final/r03_gap_trace.txt | head -30
timeout -k 10 400 python tools/bench_configs.py > $O/r03_final/r03_bench_configs.txt 2>&1; cat $O/r03_final/r03_bench_configs.txt
timeout -k 10 400 python bench.py > $O/r03_final/r03_final_bench_default.json 2> $O/r03_bench_default.err; head -c 400 $O/r03_final/r03_final_bench_default.json
