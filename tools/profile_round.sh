#!/bin/bash
# Round-end profiles on the GPU box (run from the repo root; `tools/profile_round.sh r02_final` also copies the summaries
# into profiles/ under that tag): kernel-trace statistics of the bench command, then the PMC
# passes (each its own process, never combined with a trace) over two plain forwards.  Outputs under gpurun_out/.
set -o pipefail
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out
rm -rf $O/prof_fin $O/pmcS_fin $O/pmcI_fin $O/pmcF_fin $O/pmcW_fin
( cd $R && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fin -- python3 bench.py --steps 5 --warmup 2 --cpu-crop 0 --no-f32-path > $O/prof_fin_bench.json 2> $O/prof_fin.err ) || { tail -5 $O/prof_fin.err; exit 1; }
echo "kernel trace done"
( cd $R && timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/pmcS_fin -- python3 tools/run_forward.py > $O/pmcS_fin.log 2>&1 ) || { tail -5 $O/pmcS_fin.log; exit 1; }
echo "pmc S done"
( cd $R && timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmcI_fin -- python3 tools/run_forward.py > $O/pmcI_fin.log 2>&1 ) || { tail -5 $O/pmcI_fin.log; exit 1; }
echo "pmc I done"
( cd $R && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_fin -- python3 tools/run_forward.py > $O/pmcF_fin.log 2>&1 ) || { tail -5 $O/pmcF_fin.log; exit 1; }
echo "pmc F done"
( cd $R && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_fin -- python3 tools/run_forward.py > $O/pmcW_fin.log 2>&1 ) || { tail -5 $O/pmcW_fin.log; exit 1; }
echo "pmc W done"
cd $R
# keep only what is read afterwards (the raw traces are large)
find $O/prof_fin -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py $O/pmcS_fin $O/pmcI_fin > $O/pmc_sq_fin.txt
python tools/pmc_summary.py $O/pmcF_fin $O/pmcW_fin > $O/pmc_hbm_fin.txt
python tools/make_traffic_json.py $O/pmcF_fin $O/pmcW_fin $O/hbm_traffic_fin.json HAT-S 4 720 1280 bf16
rm -rf $O/pmcS_fin $O/pmcI_fin $O/pmcF_fin $O/pmcW_fin
TAG=${1:-}
if [ -n "$TAG" ]; then
  mkdir -p $O/$TAG
  cp $(ls $O/prof_fin/*/*kernel_stats.csv | head -1) $O/$TAG/${TAG}_kernel_stats.csv
  cp $O/prof_fin_bench.json $O/$TAG/${TAG}_bench_under_rocprof.json
  cp $O/pmc_sq_fin.txt $O/$TAG/${TAG}_pmc_sq_summary.txt
  cp $O/pmc_hbm_fin.txt $O/$TAG/${TAG}_pmc_hbm_summary.txt
  cp $O/hbm_traffic_fin.json $O/$TAG/${TAG%_*}_hbm_traffic.json
fi
ls -la $O/prof_fin/*/ | head; cat $O/prof_fin_bench.json | head -c 600
