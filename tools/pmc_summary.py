#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel means."""
import csv
import re
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
        short = m.group(1) if m else name.split("(")[0].strip()
        rel = os.path.relpath(f, out).split(os.sep)[0]
        variant = " [s-scene]" if rel.endswith("_sscene") else " [s-uniform]" if rel.endswith("_suniform") else ""      # tools/pmc.sh: the fused kernel on the other synthetic captures
        acc[f"{short} @grid={row['Grid_Size']}{variant}"][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for k, ctrs in sorted(acc.items()):
    summary[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(ctrs.items())}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k, ctrs in summary.items():
    if k.startswith(("k_decode", "k_triangulate_maps", "k_scan")):
        print(k)
        for c, s in ctrs.items():
            print(f"   {c:32s} {s['mean']:16.1f}  (n={s['n']})")
