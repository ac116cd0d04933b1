#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_plan.py -m gpu -x -q -k "esc or plan" > gpurun_out/r2_t6.log 2>&1 || { tail -40 gpurun_out/r2_t6.log; exit 1; }
tail -3 gpurun_out/r2_t6.log
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "golden or headline or summaries" > gpurun_out/r2_t6b.log 2>&1 || { tail -40 gpurun_out/r2_t6b.log; exit 1; }
tail -3 gpurun_out/r2_t6b.log
HAT_BENCH_LAYERS=1 timeout -k 10 400 python bench.py --cpu-crop 0 --no-f32-path > gpurun_out/r2_bench6.json 2> gpurun_out/r2_bench6.err || { tail -20 gpurun_out/r2_bench6.err; exit 1; }
grep "^#" gpurun_out/r2_bench6.err | head -6
python - <<PY
import json
d = json.load(open("gpurun_out/r2_bench6.json"))
print(d["ms_per_step"], d["value"])
for k, v in list(d["kernels"].items())[:6]:
    print(f"  {k:55s} {v}")
PY
