#!/bin/bash
set -o pipefail
timeout -k 10 120 tools/bin/ubench_ffn2_w1 2>&1 | tee gpurun_out/r2_ub2w1_abl.log
