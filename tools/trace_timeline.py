#!/usr/bin/env python3
"""Start / end / duration of every kernel of one replayed step (dev tool). usage: trace_timeline.py <kernel_trace.csv> [from_us]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
adam = [i for i, r in enumerate(rows) if ("adam_kernel" in r["Kernel_Name"] or "adam_images_kernel" in r["Kernel_Name"])]
seg = rows[adam[-3] + 1:adam[-2] + 1]
t0 = int(seg[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"void |spv::|at::native::|\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n)[:64]
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if s / 1e3 >= lo:
        print(f"{s/1e3:8.1f} {e/1e3:8.1f} {(e-s)/1e3:7.1f}  {short(r['Kernel_Name'])}")
