#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "esc_conv13" 2>&1 | tail -2
for os in 1 0; do
HAT_ONE_STREAM=$os HAT_BENCH_LAYERS=1 timeout -k 10 400 python bench.py --cpu-crop 0 --no-f32-path > gpurun_out/r2_bench7.json 2> gpurun_out/r2_bench7.err
python - <<PY
import json
d = json.load(open("gpurun_out/r2_bench7.json"))
print("one stream=$os:", d["ms_per_step"], d["value"], {k: v["avg_ms"] for k, v in d["kernels"].items() if k in ("esc13_kernel", "cab_squeeze_kernel", "ffn2_kernel<aggr>")})
PY
done
