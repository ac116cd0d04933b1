#!/bin/bash
for v in 1 0 1 0; do
HAT_NO_N16=$v timeout -k 10 400 python bench.py --cpu-crop 0 --no-f32-path > gpurun_out/r2_bench7.json 2> gpurun_out/r2_bench7.err
python - <<PY
import json
d = json.load(open("gpurun_out/r2_bench7.json"))
print("HAT_NO_N16=$v:", d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items() if k in ("esc13_kernel", "cab_squeeze_kernel", "ffn2_kernel<aggr>")})
PY
done
