import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], " ".join("%s=%.4f" % (k.replace("conv_kernel<__bf16, ", "c<"), v["avg_ms"]) for k, v in d["kernels"].items() if "conv_kernel" in k))
