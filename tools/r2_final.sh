#!/bin/bash
# round-2 final verification on the GPU box: full GPU suite, smoke, default bench, 2-rank self-launch rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_final_tests.log 2>&1 || { tail -40 gpurun_out/r2_final_tests.log; exit 1; }
tail -2 gpurun_out/r2_final_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2_final_smoke.log 2>&1 || { tail -20 gpurun_out/r2_final_smoke.log; exit 1; }
tail -1 gpurun_out/r2_final_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err || { tail -20 gpurun_out/r2_final_bench.err; exit 1; }
HAT_BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r2_final_bench_g2.json 2> gpurun_out/r2_final_bench_g2.err || { tail -20 gpurun_out/r2_final_bench_g2.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r2_final_bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("path_f32", {}).get("ms_per_frame"), d["cpu_baseline"]["value"])
g = json.loads([l for l in open("gpurun_out/r2_final_bench_g2.json") if l.startswith("{")][-1])   # gloo logs to stdout first
print("2-rank rehearsal:", g["n_gpus"], g["scaling"], g["ms_per_step"])
PY
