#!/bin/bash
# kernel trace of a few frames; prints the idle time between consecutive tail-kernel launches of a frame
set -o pipefail
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/gaptrace
( cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaptrace -- python3 bench.py --steps 3 --warmup 2 --cpu-crop 0 --no-f32-path --no-kernel-profile > gpurun_out/gaptrace.json 2> gpurun_out/gaptrace.err ) || { tail -5 $R/gpurun_out/gaptrace.err; exit 1; }
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gaptrace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tails = [r for r in rows if "tail3_kernel" in r["Kernel_Name"] or "ffn2_kernel" in r["Kernel_Name"]]
tails = tails[-36 * 2:]          # last two frames
gaps, between = [], {}
for a, b in zip(tails[:-1], tails[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps.append(g)
gaps_hab = sorted(g for g in gaps if g < 1000)
print("tail->tail gap inside a group (us): median %.1f  p10 %.1f  p90 %.1f  n=%d" % (gaps_hab[len(gaps_hab)//2], gaps_hab[len(gaps_hab)//10], gaps_hab[len(gaps_hab)*9//10], len(gaps_hab)))
# what runs in a typical gap
a, b = tails[3], tails[4]
t0 = int(a["End_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= int(a["Start_Timestamp"]) and s <= int(b["Start_Timestamp"]):
        print("  %-50s start %+8.1f us  dur %7.1f us" % (r["Kernel_Name"][:50], (s - t0) / 1e3, (e - s) / 1e3))
# idle time of the last frame: union of kernel intervals between the first kernel after the previous frame's last tail ... crude: last frame = from
# the 36th-last tail's predecessor conv_first to the end
first = tails[-36]
fs = max(i for i, r in enumerate(rows) if int(r["Start_Timestamp"]) < int(first["Start_Timestamp"]) and "conv" in r["Kernel_Name"] and "Li1E" in r["Kernel_Name"]) if any("Li1E" in r["Kernel_Name"] for r in rows) else rows.index(first)
fr = rows[fs:]
cur_end, idle, gaps = int(fr[0]["Start_Timestamp"]), 0, []
for r in fr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        idle += s - cur_end
        gaps.append(((s - cur_end) / 1e3, r["Kernel_Name"][:40]))
    cur_end = max(cur_end, e)
span = (cur_end - int(fr[0]["Start_Timestamp"])) / 1e6
print("last frame: span %.2f ms, no kernel running for %.2f ms in %d gaps; largest:" % (span, idle / 1e6, len(gaps)), sorted(gaps, reverse=True)[:8])
PY
rm -rf gpurun_out/gaptrace
