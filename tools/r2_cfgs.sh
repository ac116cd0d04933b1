#!/bin/bash
set -o pipefail
timeout -k 10 900 python tools/bench_configs.py > gpurun_out/r2_cfgs.log 2>&1 || { tail -20 gpurun_out/r2_cfgs.log; exit 1; }
cat gpurun_out/r2_cfgs.log
HAT_GRAPH=1 timeout -k 10 300 python bench.py --cpu-crop 0 --no-f32-path --no-kernel-profile > gpurun_out/r2_bench_graph.json 2>gpurun_out/r2_bench_graph.err; python -c "
import json; d=json.load(open('gpurun_out/r2_bench_graph.json')); print('graph replay:', d['ms_per_step'], d['value'])"
