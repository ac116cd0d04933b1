"""Turn the FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, one counter per pass, tools/run_forward.py as the target)
into profiles/rNN_hbm_traffic.json: HBM bytes per launch of every hat kernel.
    python tools/make_traffic_json.py gpurun_out/pmcF gpurun_out/pmcW profiles/r01_hbm_traffic.json HAT-S 4 720 1280 bf16
Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide (16 B/lane) reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores."""
import csv, glob, json, re, sys, collections

fdir, wdir, out, model, scale, H, W, dtype = sys.argv[1:9]
BENCH_NAME = [  # rocprof kernel-name pattern -> the name bench.py prints
    (r"tail3l_kernel", "tail3l_kernel"), (r"tail3_kernel", "tail3_kernel"),
    (r"ffn2_kernelILi0ELb1E|ffn2_kernel<0, true", "ffn2_kernel<aggr>"), (r"ffn2_kernel", "ffn2_kernel"),
    (r"esc13_kernel", "esc13_kernel"), (r"ffn_kernel", "ffn_kernel<__bf16>"), (r"ocab_attn", "ocab_attn_kernel<__bf16>"), (r"aggr_cab_kernel", "aggr_cab_kernel"),
    (r"pw_kernel.*ELi5E", "pw_kernel<__bf16, 9, 5>"), (r"pw_kernel.*ELi9E", "pw_kernel<__bf16, 9, 9>"),
    (r"cab_squeeze_kernel(ILi5E|<5)", "cab_squeeze_kernel"), (r"cab_squeeze_kernel(ILi2E|<2)", "cab_squeeze_kernel<2, planes>"),
    (r"tap3_kernel.*41, 144", "tap3_kernel<__bf16, 1>"), (r"tap3_kernel.*ELi3ELi8E", "tap3_kernel<__bf16, 9>"),
    (r"conv_kernelIDF16bLi8ELi2ELi9E", "conv_kernel<__bf16, 8, 2, 9>"), (r"conv_kernelIDF16bLi8ELi2ELi1E", "conv_kernel<__bf16, 8, 2, 1>"),
    (r"conv_kernelIDF16bLi8ELi2ELi4E", "conv_kernel<__bf16, 8, 2, 4>"), (r"ln_kernel", "ln_kernel"),
]


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fe, wr = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
kernels = {}
for k in fe:
    for pat, name in BENCH_NAME:
        if re.search(pat, k):
            e = kernels.setdefault(name, {"rocprof_names": [], "launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
            e["rocprof_names"].append(k[:120])
            e["launches"] += len(fe[k])
            e["fetch_kib"] += sum(fe[k])
            e["write_kib"] += sum(wr.get(k, []))
            break
for e in kernels.values():
    n = e.pop("launches")
    f, w = e.pop("fetch_kib") / n, e.pop("write_kib") / n
    e.update({"launches_sampled": n, "FETCH_SIZE_KiB_per_launch": round(f, 1), "WRITE_SIZE_KiB_per_launch": round(w, 1),
              "hbm_read_bytes_per_launch": round(2 * f * 1024), "hbm_write_bytes_per_launch": round(w * 1024),
              "hbm_bytes_per_launch": round((2 * f + w) * 1024)})
json.dump({"workload": [model, int(scale), int(H), int(W), dtype],
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/run_forward.py (two forwards); "
                     "mean per launch; FETCH_SIZE doubled (gfx950 128-byte requests tallied at 64 B), KiB -> bytes",
           "kernels": kernels}, open(out, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in kernels.items()}, indent=1))
