#!/bin/bash
# Round-end verification on the GPU box: GPU tests, smoke, two default bench runs.
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -30 gpurun_out/final_tests.log; exit 1; }
tail -3 gpurun_out/final_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1 || { tail -20 gpurun_out/final_smoke.log; exit 1; }
tail -1 gpurun_out/final_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/final_bench1.json 2> gpurun_out/final_bench1.err || { tail -20 gpurun_out/final_bench1.err; exit 1; }
timeout -k 10 400 python bench.py --cpu-crop 0 > gpurun_out/final_bench2.json 2> gpurun_out/final_bench2.err || exit 1
python - <<PY
import json
for f in ("gpurun_out/final_bench1.json", "gpurun_out/final_bench2.json"):
    d = json.load(open(f))
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("cpu_baseline"))
PY
