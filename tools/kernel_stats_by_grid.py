#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) statistics.

`rocprofv3 --stats` groups by kernel name only; bench.py launches the same kernels on two workloads in one run (the
4096x3000 scan and, for the throughput-mode extra, 1920x1080 scans), so the averages have to be split by grid size to be
compared with the HIP-event averages bench.py prints.  usage: kernel_stats_by_grid.py <kernel_trace.csv> <out.csv>"""
import csv
import re
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for r in csv.DictReader(open(src)):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"].split("(")[0]
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    acc[(name, grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for (name, grid, wg), v in acc.items():
    v.sort()
    rows.append((sum(v), name, grid, wg, len(v), sum(v) / len(v), v[0], v[len(v) // 2], v[-1]))
rows.sort(reverse=True)
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Grid_Size(threads)", "Workgroup_Size", "Calls", "TotalNs", "AverageNs", "MinNs", "MedianNs", "MaxNs"])
    for tot, name, grid, wg, n, avg, mn, med, mx in rows:
        w.writerow([name, grid, wg, n, tot, f"{avg:.1f}", mn, med, mx])
        print(f"{name:48s} grid {grid:10d} wg {wg:4d} calls {n:5d} avg {avg / 1e3:9.1f} us  median {med / 1e3:9.1f} us")
