#!/usr/bin/env python3
"""gpurun_out/final/{kt,pmc,bench_*.json} -> profiles/r01_* (kernel stats, per-grid stats, PMC summary, traffic.json)."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "final")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
shutil.copy(max(glob.glob(os.path.join(src, "kt", "*", "*kernel_stats.csv")), key=os.path.getmtime), os.path.join(dst, f"{tag}_kernel_stats_bench_default.csv"))
print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_stats_by_grid.py"), max(glob.glob(os.path.join(src, "kt", "*", "*kernel_trace.csv")), key=os.path.getmtime),
                      os.path.join(dst, f"{tag}_kernel_stats_by_grid.csv")], capture_output=True, text=True).stdout.split("k_synth")[0])
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(src, "pmc")], capture_output=True)
shutil.copy(os.path.join(src, "pmc", "pmc_summary.json"), os.path.join(dst, f"{tag}_pmc_summary_c3.json"))
d = json.load(open(os.path.join(dst, f"{tag}_pmc_summary_c3.json")))
dec = [k for k in d if k.startswith("k_decode_pk") and "false> @grid=3072000" in k][0]
fus = [k for k in d if k.startswith("k_decode_pk") and "true> @grid=3072000" in k][0]
traffic = lambda k: int(round((2 * d[k]["FETCH_SIZE"]["mean"] + d[k]["WRITE_SIZE"]["mean"]) * 1024))   # noqa: E731
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh), profiles/%s_pmc_summary_c3.json; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
        "gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)" % tag)
t = {"c3_4096x3000x44/g1/split": {"kernel": dec, "hbm_bytes_per_launch": traffic(dec), "fetch_size_kb": d[dec]["FETCH_SIZE"]["mean"],
                                  "write_size_kb": d[dec]["WRITE_SIZE"]["mean"], "source": note},
     "c3_4096x3000x44/g1/fused": {"kernel": fus, "hbm_bytes_per_launch": traffic(fus), "fetch_size_kb": d[fus]["FETCH_SIZE"]["mean"],
                                  "write_size_kb": d[fus]["WRITE_SIZE"]["mean"],
                                  "source": note + "; includes the camera-ray table (98 MB) and projector-ray gathers, which are not algorithmic bytes"}}
json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, f"{tag}_bench_default.json"))
line = [l for l in open(os.path.join(src, "bench_under_rocprof.log")) if l.startswith("{")][0]
open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(line)
b, p = json.load(open(os.path.join(dst, f"{tag}_bench_default.json"))), json.loads(line)
for name, j in (("plain", b), ("under rocprof", p)):
    print(f"{name:14s} value {j['value']:9.1f}  fused kernel {j['roofline']['avg_launch_ms'] * 1e3:6.1f} us frac {j['roofline']['frac']:.3f} | split decode "
          f"{j['split_pipeline']['roofline']['avg_launch_ms'] * 1e3:6.1f} us frac {j['split_pipeline']['roofline']['frac']:.3f} | decode alone "
          f"{j['decode_kernel_alone']['roofline']['frac']:.3f} | throughput {j['throughput_mode']['value']:9.1f}")
print({k: v["hbm_bytes_per_launch"] for k, v in t.items()})
