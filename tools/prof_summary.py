#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV into a short table (names truncated).
usage: prof_summary.py <kernel_stats.csv> [n_steps_for_per_step_columns] [top]"""
import csv, re, sys

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(path)))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"at::native::|at::cuda::|void ", "", n)
    m = re.match(r"(spv::\w+<[^>]*>+|spv::\w+|Cijk_\w{0,30}|[\w:]+)", n)
    base = m.group(1) if m else n
    if "elementwise" in base or "kernelPointwise" in base or "reduce_kernel" in base:
        f = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|\w+_cuda_kernel|MeanOps|sum_functor|and_kernel|FillFunctor|direct_copy\w*)", n)
        base += "[" + (f.group(1) if f else "?") + "]"
    return base[:90]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {path}\n# total kernel time {tot/1e6:.3f} ms over {steps:g} steps = {tot/1e6/steps:.3f} ms/step; launches/step {sum(int(r['Calls']) for r in rows)/steps:.0f}")
print(f"{'kernel':92s} {'calls/step':>10s} {'avg_us':>10s} {'ms/step':>9s} {'%':>6s}")
for r in rows[:top]:
    c, t = int(r["Calls"]), float(r["TotalDurationNs"])
    print(f"{short(r['Name']):92s} {c/steps:10.1f} {float(r['AverageNs'])/1e3:10.1f} {t/1e6/steps:9.3f} {100*t/tot:6.2f}")
