#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 --pmc CSV output. usage: pmc_summary.py <counter_collection.csv> <kernel-substring>"""
import csv, sys, collections
path, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
