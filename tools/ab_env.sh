#!/bin/bash
# A/B on ONE box: default bench with and without an environment switch, twice each, interleaved.  usage: tools/ab_env.sh VAR=VALUE
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
  for arm in base "$1"; do
    if [ "$arm" = base ]; then
      timeout -k 10 300 python bench.py --no-f32-path --cpu-crop 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -20 gpurun_out/ab.err; exit 1; }
    else
      env "$arm" timeout -k 10 300 python bench.py --no-f32-path --cpu-crop 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -20 gpurun_out/ab.err; exit 1; }
    fi
    python - "$arm" <<PY
import json, sys
d = json.loads([l for l in open("gpurun_out/ab.json") if l.startswith("{")][-1])
print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], flush=True)
PY
  done
done
