#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_plan.py -m gpu -x -q -k "hatx or golden or plan" > gpurun_out/r2_t6.log 2>&1 || { tail -40 gpurun_out/r2_t6.log; exit 1; }
tail -3 gpurun_out/r2_t6.log
