#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "fused_ffn" > gpurun_out/r2_t2.log 2>&1 || { tail -40 gpurun_out/r2_t2.log; exit 1; }
tail -3 gpurun_out/r2_t2.log
timeout -k 10 120 tools/bin/ubench_ffn2 > gpurun_out/r2_ub2.log 2>&1 || { tail gpurun_out/r2_ub2.log; exit 1; }
cat gpurun_out/r2_ub2.log
UB_PHASES=1 timeout -k 10 120 tools/bin/ubench_ffn2 > gpurun_out/r2_ub2p.log 2>&1 || { tail gpurun_out/r2_ub2p.log; exit 1; }
cat gpurun_out/r2_ub2p.log
UB_ONLY_FULL=1 timeout -k 10 120 tools/bin/ubench_ffn > gpurun_out/r2_ub1.log 2>&1; cat gpurun_out/r2_ub1.log
