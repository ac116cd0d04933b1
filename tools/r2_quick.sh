#!/bin/bash
# quick GPU check after a tail-kernel change: its parity tests, then the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "ffn2 or hab_tail or headline or tiny or cfg or ocab or plan or linear or conv" > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
tail -2 gpurun_out/quick_tests.log
timeout -k 10 400 python bench.py --no-f32-path > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || { tail -20 gpurun_out/quick_bench.err; exit 1; }
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/quick_bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("avg_ms"))
PY
