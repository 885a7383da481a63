#!/usr/bin/env python3
"""Mean of every counter per (kernel, grid) over all rocprofv3 --pmc passes under a directory.  usage: pmc_by_kernel.py <dir> [substring ...]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
want = sys.argv[2:] or ["k_decode_pk", "k_triangulate_maps_lds", "k_xmajor"]
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(w in r["Kernel_Name"] for w in want):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[(name, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(os.path.relpath(f, root).split(os.sep)[0])
    for (name, grid, ctr), v in sorted(agg.items()):
        print(f"   {name:52s} grid {grid:9d}  {ctr:24s} n {len(v):3d}  mean {sum(v) / len(v):12.0f}")
