#!/bin/bash
# full GPU verification: tests (incl. the reference-pinned full-size ones), smoke, default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r2_tests_full.log 2>&1 || { tail -40 gpurun_out/r2_tests_full.log; exit 1; }
tail -14 gpurun_out/r2_tests_full.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2_smoke.log 2>&1 || { tail -20 gpurun_out/r2_smoke.log; exit 1; }
tail -1 gpurun_out/r2_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err || { tail -20 gpurun_out/r2_bench2.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r2_bench2.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("path_f32"))
for k, v in list(d["kernels"].items())[:14]:
    print(f"  {k:55s} {v}")
PY
