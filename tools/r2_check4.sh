#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "hab_tail or fused_ffn2 or folded_cab" > gpurun_out/r2_t4.log 2>&1 || { tail -40 gpurun_out/r2_t4.log; exit 1; }
tail -3 gpurun_out/r2_t4.log
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "golden or headline or summaries" > gpurun_out/r2_t4b.log 2>&1 || { tail -40 gpurun_out/r2_t4b.log; exit 1; }
tail -3 gpurun_out/r2_t4b.log
timeout -k 10 400 python bench.py --cpu-crop 0 --no-f32-path > gpurun_out/r2_bench4.json 2> gpurun_out/r2_bench4.err || { tail -20 gpurun_out/r2_bench4.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r2_bench4.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
for k, v in list(d["kernels"].items())[:8]:
    print(f"  {k:55s} {v}")
PY
