#!/bin/bash
set -o pipefail
HAT_BENCH_LAYERS=1 timeout -k 10 400 python bench.py --cpu-crop 0 --no-f32-path > gpurun_out/r2_bench5.json 2> gpurun_out/r2_bench5.err || { tail -20 gpurun_out/r2_bench5.err; exit 1; }
grep "^#" gpurun_out/r2_bench5.err | head -40
python - <<PY
import json
d = json.load(open("gpurun_out/r2_bench5.json"))
print(d["ms_per_step"], d["value"])
for k, v in list(d["kernels"].items()):
    print(f"  {k:55s} {v}")
PY
