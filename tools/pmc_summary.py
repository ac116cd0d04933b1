"""Aggregate rocprofv3 counter_collection CSVs: mean counter value per dispatch for every hat kernel.
    python tools/pmc_summary.py gpurun_out/pmcA [gpurun_out/pmcB ...]"""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "hat" not in k and "_kernel" not in k or "at::native" in k:
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", k)
            k = re.sub(r"\(.*", "", k).replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
for k in sorted(acc):
    print(f"== {k}  vgpr/agpr/lds/scratch/grid/wg = {meta[k]}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
