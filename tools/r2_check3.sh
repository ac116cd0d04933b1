#!/bin/bash
set -o pipefail
echo "== warm-up on"; timeout -k 10 120 tools/bin/ubench_ffn2 2>&1 | head -4
echo "== warm-up off"; timeout -k 10 120 tools/bin/ubench_ffn2_nopf 2>&1 | head -4
UB_PHASES=1 UB_AGGR=1 timeout -k 10 120 tools/bin/ubench_ffn2 2>&1 | head -4
