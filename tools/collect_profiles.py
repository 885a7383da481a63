#!/usr/bin/env python3
"""gpurun_out/final/{kt,pmc,bench_*.json} -> profiles/<tag>_* (kernel stats, per-grid stats, PMC summary + the raw counter rows of the
bench kernels, traffic.json stamped with the fingerprint of the kernel sources it was measured on).  usage: collect_profiles.py r03"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src = os.path.join(ROOT, "gpurun_out", "final")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
C3_GRID = 4096 * 3000 // 4          # threads of a 4-pixels-per-lane kernel over 4096x3000


def newest(pattern):
    return max(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)


shutil.copy(newest("kt/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_bench_default.csv"))
print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_stats_by_grid.py"), newest("kt/**/*kernel_trace.csv"),
                      os.path.join(dst, f"{tag}_kernel_stats_by_grid.csv")], capture_output=True, text=True).stdout.split("k_synth")[0])
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(src, "pmc")], capture_output=True)
shutil.copy(os.path.join(src, "pmc", "pmc_summary.json"), os.path.join(dst, f"{tag}_pmc_summary_c3.json"))
# raw counter rows of the scan kernels at the benchmark grid (what the summary and traffic.json are computed from)
with open(os.path.join(dst, f"{tag}_pmc_raw_c3.csv"), "w", newline="") as out:
    w = None
    for f in sorted(glob.glob(os.path.join(src, "pmc", "pass*", "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if ("k_decode_pk" in row["Kernel_Name"] or "k_triangulate_maps_lds" in row["Kernel_Name"]) and int(row["Grid_Size"]) == C3_GRID:
                if w is None:
                    w = csv.DictWriter(out, fieldnames=["pass"] + list(row.keys()))
                    w.writeheader()
                w.writerow({"pass": os.path.relpath(f, os.path.join(src, "pmc")).split(os.sep)[0], **row})
d = json.load(open(os.path.join(dst, f"{tag}_pmc_summary_c3.json")))


def kernel_key(fused, grid=C3_GRID, variant="", bgr=False):
    """k_decode_pk<PX, BLOCK, NT, MULTI, ABL, FUSE, NS, BGR> @grid [variant]: FUSE = 0 is the decode kernel, 1 / 2 / 3 the fused scan kernel, BGR = 1 its BGR-reading form."""
    for k in d:
        m = re.match(r"k_decode_pk<([^>]*)> @grid=(\d+)( \[[a-z-]+\])?$", k)
        if m and int(m.group(2)) == grid and (m.group(3) or "").strip() == variant and "FETCH_SIZE" in d[k] and "WRITE_SIZE" in d[k]:
            args = [a.strip() for a in m.group(1).split(",")]
            if (int(args[5]) != 0) == fused and ((len(args) > 7 and args[7] == "1") == bgr):
                return k
    return None


traffic = lambda k: int(round((2 * d[k]["FETCH_SIZE"]["mean"] + d[k]["WRITE_SIZE"]["mean"]) * 1024))   # noqa: E731
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh), profiles/%s_pmc_summary_c3.json + %s_pmc_raw_c3.csv; bytes = "
        "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)" % (tag, tag))
fp = bench.csrc_fingerprint()
C2_GRID = 1920 * 1080 // 4
t = {}
# key = workload / g1 / pipeline / scene (bench.py: roofline.traffic falls back to these when it could not collect its own counters)
for key, k in (("c3_4096x3000x44/g1/split/s-scene", kernel_key(False)), ("c3_4096x3000x44/g1/fused/physical", kernel_key(True)),
               ("c3_4096x3000x44/g1/fused/s-scene", kernel_key(True, variant="[s-scene]")), ("c3_4096x3000x44/g1/fused/s-uniform", kernel_key(True, variant="[s-uniform]")),
               ("c2_1920x1080x44/g1/fused/physical", kernel_key(True, grid=C2_GRID)), ("c3_4096x3000x44/g1/fused-bgr/physical", kernel_key(True, bgr=True))):
    if k is None:
        print("no PMC rows for", key)
        continue
    t[key] = {"kernel": k, "hbm_bytes_per_launch": traffic(k), "fetch_size_kb": d[k]["FETCH_SIZE"]["mean"], "write_size_kb": d[k]["WRITE_SIZE"]["mean"],
              "csrc_fingerprint": fp, "source": note}
json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, f"{tag}_bench_default.json"))
for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "line_*.json")):       # full reports (bench_*) and the printed lines (line_*)
    if os.path.basename(f) not in ("bench_default.json",) and not re.search(r"bench_steps20_run[2-5]", f):
        shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(f)}"))
if os.path.exists(os.path.join(src, "driver_runs.txt")):
    shutil.copy(os.path.join(src, "driver_runs.txt"), os.path.join(dst, f"{tag}_driver_runs.txt"))
    with open(os.path.join(dst, f"{tag}_driver_runs.txt"), "a") as fo:                       # mean vs median of every leg of the five runs
        for i in range(1, 6):
            try:
                jj = json.load(open(os.path.join(src, f"bench_steps20_run{i}.json")))
            except OSError:
                continue
            legs = {"headline": jj["roofline"]}
            legs.update({"scene " + k: v["roofline"] for k, v in jj.get("scenes", {}).items() if "roofline" in v})
            if "decode_kernel_headline" in jj:
                legs["decode kernel"] = jj["decode_kernel_headline"]["roofline"]
            fo.write(f"run {i}: value {jj['value']} " + "; ".join(f"{k}: frac {r['frac']} mean {r['frac_mean']} outliers {r.get('outliers')}" for k, r in legs.items()) +
                     f" | seconds {jj.get('leg_seconds')}\n")
b, p = json.load(open(os.path.join(dst, f"{tag}_bench_default.json"))), json.load(open(os.path.join(src, "bench_under_rocprof.json")))
for name, j in (("plain", b), ("under rocprof", p)):
    print(f"{name:14s} value {j['value']:9.1f}  fused kernel {j['roofline']['avg_launch_ms'] * 1e3:6.1f} us frac {j['roofline']['frac']:.3f} | split decode "
          f"{j['split_pipeline']['roofline']['avg_launch_ms'] * 1e3:6.1f} us frac {j['split_pipeline']['roofline']['frac']:.3f} | decode alone "
          f"{j['decode_kernel_alone']['roofline']['frac']:.3f} | throughput {j.get('throughput_mode', {}).get('value')}")
alg = {"split": 48, "fused": 56, "fused-bgr": 144}
print({k: (v["hbm_bytes_per_launch"], round(v["hbm_bytes_per_launch"] / (alg[k.split("/")[2]] * (1920 * 1080 if k.startswith("c2") else 4096 * 3000)), 3)) for k, v in t.items()},
      "fingerprint", fp)

# the "next" rows, the list stage, the host API and the store-pattern microbenchmark of the same box
try:
    print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_stats_by_grid.py"), newest("kt_c2/**/*kernel_trace.csv"),
                          os.path.join(dst, f"{tag}_kernel_stats_by_grid_c2_1920x1080x44.csv")], capture_output=True, text=True).stdout.split("k_synth")[0])
except (ValueError, IndexError, OSError) as e_:
    print("no 1920x1080 kernel trace", e_)
for sub, name in (("kt_next", "kernel_stats_next_rows"), ("kt_lists", "kernel_stats_list_stage")):
    try:
        shutil.copy(newest(sub + "/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_{name}.csv"))
    except ValueError:
        print("no", sub)
for f, name in (("next_rows.log", "next_rows.txt"), ("lists_plain.log", "list_stage.txt"), ("host_api.log", "host_api.txt"), ("dropin.log", "dropin.txt"), ("write_patterns.txt", "write_patterns.txt"),
                ("cloud.log", "reference_product.txt"), ("pmc_lists/summary.txt", "pmc_list_stage.txt"), ("stream_rates.txt", "stream_rates.txt"),
                ("ingest.log", "ingest.txt"), ("batch_rounds.log", "batch_rounds.txt"), ("ab_guard.log", "guard_forms.txt"), ("tiled_layout.log", "tiled_layout.txt")):
    if os.path.exists(os.path.join(src, f)):
        keep = [ln for ln in open(os.path.join(src, f), errors="replace") if not re.match(r"^(RCCL|HIP|ROCm|Hostname|Librccl|[WEI]\d{8}) ", ln)]
        open(os.path.join(dst, f"{tag}_{name}"), "w").writelines(keep)
