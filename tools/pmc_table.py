#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters across several pass directories (dev tool)."""
import csv, sys, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not ("spv" in k or "adam" in k or "pack" in k or "fc1" in k):
                continue
            k = re.sub(r"spv::|void |\(.*\)$", "", k).replace("gemm_kernel<GemmCfg<", "gemm<")[:64]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(66) + " ".join(n.replace("SQ_", "")[:14].rjust(15) for n in names))
rows = []
for k, cs in acc.items():
    rows.append((-(sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(len(cs.get("SQ_WAVE_CYCLES", [1])), 1)), k, cs))
for _, k, cs in sorted(rows):
    print(k.ljust(66) + " ".join((f"{sum(cs[n])/len(cs[n]):15.0f}" if n in cs else " " * 15) for n in names))
