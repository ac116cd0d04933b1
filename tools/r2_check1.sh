#!/bin/bash
# round 2, first GPU call: issue-cost microbenchmarks, the GPU suite without the full-size fixtures, self-launch rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 tools/bin/ubench_valu > gpurun_out/r2_ubench_valu.log 2>&1 || { tail gpurun_out/r2_ubench_valu.log; exit 1; }
cat gpurun_out/r2_ubench_valu.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not vs_reference and not cfg4" > gpurun_out/r2_tests1.log 2>&1 || { tail -40 gpurun_out/r2_tests1.log; exit 1; }
tail -3 gpurun_out/r2_tests1.log
HAT_BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r2_bench_g2.json 2> gpurun_out/r2_bench_g2.err || { tail -20 gpurun_out/r2_bench_g2.err; exit 1; }
cat gpurun_out/r2_bench_g2.json | cut -c1-600
timeout -k 10 400 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err || { tail -20 gpurun_out/r2_bench1.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r2_bench1.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("path_f32"), d.get("cpu_baseline"))
PY
