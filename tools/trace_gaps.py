#!/usr/bin/env python3
"""Busy / idle breakdown of one replayed step from a rocprofv3 kernel trace (dev tool). usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if ("adam_kernel" in r["Kernel_Name"] or "adam_images_kernel" in r["Kernel_Name"])]
a, b = adam[-3], adam[-2]          # one full step between two optimiser launches
seg = rows[a + 1:b + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e in iv)
print(f"kernels {len(seg)}  wall {1e-3*(t1-t0):.1f} us  busy(union) {1e-3*busy:.1f} us  sum of durations {1e-3*tot:.1f} us  idle {1e-3*(t1-t0-busy):.1f} us in {len(gaps)} gaps")
def short(n):
    return re.sub(r"void |spv::|at::native::|\(.*", "", n)[:60]
print("largest gaps (us, after kernel):")
ends = {int(r["End_Timestamp"]): r["Kernel_Name"] for r in seg}
for g, at in sorted(gaps, reverse=True)[:12]:
    print(f"  {g*1e-3:7.1f}  after {short(ends.get(at, '?'))}")
