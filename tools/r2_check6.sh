#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_plan.py tests/test_gpu_ops.py -m gpu -x -q -k "golden or headline or summaries or plan or hab_tail or ffn2" > gpurun_out/r2_t6.log 2>&1 || { tail -40 gpurun_out/r2_t6.log; exit 1; }
tail -3 gpurun_out/r2_t6.log
for i in 1 2; do timeout -k 10 400 python bench.py --cpu-crop 0 --no-f32-path > gpurun_out/r2_bench6.json 2> gpurun_out/r2_bench6.err || { tail -20 gpurun_out/r2_bench6.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r2_bench6.json"))
print(d["ms_per_step"], d["value"], {k: v["avg_ms"] for k, v in d["kernels"].items() if k in ("esc13_kernel", "cab_squeeze_kernel", "ffn2_kernel<aggr>")})
PY
done
