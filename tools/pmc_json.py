#!/usr/bin/env python3
"""Per-kernel, per-launch averages of every counter of a set of rocprofv3 --pmc passes (separate passes, --kernel-trace only), as JSON
+ a text table.  usage: pmc_json.py "<workload tag>" <steps of each pass> <out.json> <out.txt> <pass dir> [<pass dir> ...]

The workload tag is what bench.py matches before it attaches counters to a line (`roofline.traffic`, `decoder_chain`):
"<preset> <precision> <count dtype> B<batch> G<genes>".  Units as rocprofv3 prints them: FETCH_SIZE / WRITE_SIZE in KiB; SQ_* in
quad-cycles (SQ_VALU_MFMA_BUSY_CYCLES in cycles) summed over all SIMDs; GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs."""
import collections
import csv
import glob
import json
import re
import sys

tag, steps, out_json, out_txt = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[5:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not ("spv" in k or "adam" in k or "fc1" in k or "prepare_log1p" in k):
                continue
            k = re.sub(r"^void |spv::", "", k)
            k = re.sub(r"\(.*\)$", "", k)
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {}
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    e = {"launches_per_step": n / steps}
    for c, v in cs.items():
        e[c] = sum(v) / len(v)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_frac"] = (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (e["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_ACTIVE_INST_VALU" in e and e.get("GRBM_GUI_ACTIVE"):
        e["valu_busy_frac"] = (e["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / (e["GRBM_GUI_ACTIVE"] / 8.0)
    kern[k] = e
json.dump({"workload": tag, "steps_per_pass": steps, "kernels": kern}, open(out_json, "w"), indent=1, sort_keys=True)
cols = ["launches_per_step", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_frac", "SQ_ACTIVE_INST_VALU", "valu_busy_frac",
        "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT"]
with open(out_txt, "w") as fh:
    fh.write(f"# workload: {tag}\n# per launch; FETCH_SIZE / WRITE_SIZE KiB (raw), GRBM_GUI_ACTIVE cycles over 8 XCDs, SQ_* summed over 1024 SIMDs\n")
    fh.write("kernel".ljust(72) + " ".join(c.replace("SQ_", "")[:16].rjust(17) for c in cols) + "\n")
    for k, e in sorted(kern.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches_per_step"]):
        fh.write(k[:70].ljust(72) + " ".join((f"{e[c]:17.3f}" if c.endswith("frac") or c == "launches_per_step" else f"{e[c]:17.0f}") if c in e else " " * 17 for c in cols) + "\n")
