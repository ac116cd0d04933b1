import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"])
    for k, v in d["kernels"].items():
        if any(s in k for s in ("esc", "cab", "tail3")): print("   %-40s %3d x %.4f" % (k, v["launches_per_step"], v["avg_ms"]))
