#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -m gpu -x -q > gpurun_out/r2_t6.log 2>&1 || { tail -40 gpurun_out/r2_t6.log; exit 1; }
tail -3 gpurun_out/r2_t6.log
