#!/usr/bin/env python3
"""Prints the "numbers of this run" of profiles/README.md from the files tools/collect_profiles.py wrote.  usage: profile_table.py r03"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_{n}")          # noqa: E731
j, u = json.load(open(P("bench_default.json"))), json.load(open(P("bench_under_rocprof.json")))
c2 = json.load(open(P("bench_c2_1920x1080x44.json")))
g = {(x["Kernel"], x["Grid_Size(threads)"]): x for x in csv.DictReader(open(P("kernel_stats_by_grid.csv")))}
g2 = {(x["Kernel"], x["Grid_Size(threads)"]): x for x in csv.DictReader(open(P("kernel_stats_by_grid_c2_1920x1080x44.csv")))}
us = lambda row, col="AverageNs": float(row[col]) / 1e3   # noqa: E731


def gk(tab, name, grid):
    return next(v for (k, gr), v in tab.items() if k.startswith(name) and gr == grid)


r, o = j["roofline"], j["other_scene"]
fz, dz, tz = gk(g, "k_decode_pk<4, 128, 1, false, 0, 3, 44, 0>", "3072000"), gk(g, "k_decode_pk<4, 128, 1, false, 0, 0, 44, 0>", "3072000"), gk(g, "k_triangulate_maps_lds<1>", "3072000")
print(f"fused 4096x3000x44 (physical): events avg {r['avg_launch_ms'] * 1e3:.1f} us median {r['median_launch_ms'] * 1e3:.1f} min {r['min_launch_ms'] * 1e3:.1f} p95 {r['p95_launch_ms'] * 1e3:.1f} "
      f"-> {j['value']:.0f} Mpix/s frac {r['frac']:.3f} ({r['frac_incl_maps']:.3f} incl. maps); S-scene same run {o['roofline']['avg_launch_ms'] * 1e3:.1f} us frac {o['roofline']['frac']:.3f}; "
      f"profiled run {u['roofline']['avg_launch_ms'] * 1e3:.1f} us | rocprof avg {us(fz):.1f} median {us(fz, 'MedianNs'):.1f} over {fz['Calls']} launches")
sp = j["split_pipeline"]["roofline"]
print(f"decode kernel: split {sp['avg_launch_ms'] * 1e3:.1f} us frac {sp['frac']:.3f}, alone {j['decode_kernel_alone']['roofline']['frac']:.3f} | rocprof avg {us(dz):.1f} median {us(dz, 'MedianNs'):.1f} ({dz['Calls']})")
print(f"dense triangulation kernel: rocprof {us(tz):.1f} us ({tz['Calls']})")
r2, o2 = c2["roofline"], c2["other_scene"]
f2 = gk(g2, "k_decode_pk<4, 128, 1, false, 0, 3, 44, 0>", "518400")
print(f"fused 1920x1080x44 (physical): events avg {r2['avg_launch_ms'] * 1e3:.2f} us -> {c2['value']:.0f} Mpix/s frac {r2['frac']:.3f}; S-scene {o2['roofline']['avg_launch_ms'] * 1e3:.2f} us frac "
      f"{o2['roofline']['frac']:.3f} (guard {o2['guard_flagged_pixels']}) | rocprof avg {us(f2):.1f} median {us(f2, 'MedianNs'):.1f} over {f2['Calls']} launches (both scenes, sustained leg)")
rp = j["reference_product"]
ls = {x["Name"]: x for x in csv.DictReader(open(P("kernel_stats_list_stage.csv")))}
find = lambda d, key: next(v for k, v in d.items() if key in k)   # noqa: E731
sc, ct, pf, cs = find(ls, "xmajor_lines" if any("xmajor_lines" in k for k in ls) else "xmajor_scatter"), find(ls, "xmajor_count"), find(ls, "colprefix"), find(ls, "colscan")
print(f"reference product: {rp['ms_per_scan']:.3f} ms per scan = {rp['value']:.0f} Mpix/s (dense route {rp['via_dense_xyz']['ms_per_scan']:.3f} ms); list stage {rp['list_stage_ms']:.3f} ms, "
      f"{rp['list_stage_bytes'] / 1e6:.0f} MB -> {rp['list_stage_roofline']['frac']:.2f}, {rp['executed'].get('list_kernel')} | rocprof scatter {us(sc):.1f} (min {us(sc, 'MinNs'):.1f}) count {us(ct):.1f} prefix {us(pf):.1f} scan {us(cs):.1f} us ({sc['Calls']} builds)")
nx = {x["Name"]: x for x in csv.DictReader(open(P("kernel_stats_next_rows.csv")))}
fd, bg = find(nx, "frame_diff"), find(nx, "bgr")
print(f"next rows: k_frame_diff_u8x16 {us(fd):.1f} us ({541e6 / float(fd['AverageNs']) * 1e9 / 8e12:.2f}), k_bgr_to_gray {us(bg):.1f} us ({196.6e6 / float(bg['AverageNs']) * 1e9 / 8e12:.2f})")
pm = json.load(open(P("pmc_summary_c3.json")))
for label, key in (("fused physical", "k_decode_pk<4, 128, 1, false, 0, 3, 44, 0> @grid=3072000"), ("fused S-scene", "k_decode_pk<4, 128, 1, false, 0, 3, 44, 0> @grid=3072000 [s-scene]"),
                   ("fused S-uniform", "k_decode_pk<4, 128, 1, false, 0, 3, 44, 0> @grid=3072000 [s-uniform]"), ("fused from BGR", "k_decode_pk<4, 128, 1, false, 0, 3, 44, 1> @grid=3072000"),
                   ("decode", "k_decode_pk<4, 128, 1, false, 0, 0, 44, 0> @grid=3072000"), ("dense tri", "k_triangulate_maps_lds<1> @grid=3072000")):
    k = next((x for x in pm if x == key), None)
    if not k or "FETCH_SIZE" not in pm[k]:
        print("PMC", label, "missing")
        continue
    F, Wr = pm[k]["FETCH_SIZE"]["mean"], pm[k]["WRITE_SIZE"]["mean"]
    extra = ""
    if "SQ_ACTIVE_INST_VALU" in pm[k] and "GRBM_GUI_ACTIVE" in pm[k]:
        va, gui = pm[k]["SQ_ACTIVE_INST_VALU"]["mean"] * 4 / 1024, pm[k]["GRBM_GUI_ACTIVE"]["mean"] / 8
        extra = f"; VALU busy {va / 1e3:.0f} k of {gui / 1e3:.0f} k cycles = {va / gui:.2f}"
    if "SQ_LDS_BANK_CONFLICT" in pm[k] and "SQ_LDS_IDX_ACTIVE" in pm[k]:
        extra += f"; LDS bank-conflict cycles {pm[k]['SQ_LDS_BANK_CONFLICT']['mean']:.0f} of {pm[k]['SQ_LDS_IDX_ACTIVE']['mean'] / 1e6:.2f} M LDS-array cycles"
    print(f"PMC {label}: FETCH_SIZE {F:.0f} KB x2 + WRITE_SIZE {Wr:.0f} KB = {(2 * F + Wr) * 1024 / 1e6:.1f} MB per launch (n={pm[k]['FETCH_SIZE']['n']}){extra}")
for name, row in j.get("scenes", {}).items():
    print(f"scene {name}: fused {row['avg_launch_ms'] * 1e3:.1f} us frac {row['frac']:.3f} ({row['fused_time_over_s_scene']} x S-scene), valid {row['valid_pixels_per_scan']}, flagged {row['guard_flagged_pixels']}, "
          f"decode kernel {row['decode_kernel']['avg_launch_ms'] * 1e3:.1f} us frac {row['decode_kernel']['frac']:.3f}")
tr, ig = j["two_runs"], j["ingest"]
print(f"two runs: {tr['roofline']['avg_launch_ms'] * 1e3:.1f} us frac {tr['roofline']['frac']:.3f}; from BGR: {ig['ingest_fused']['roofline']['avg_launch_ms'] * 1e3:.1f} us frac "
      f"{ig['ingest_fused']['roofline']['frac']:.3f}, per scan {ig['ingest_fused']['ms_per_step']:.4f} vs {ig['ingest_separate']['ms_per_step']:.4f} ms ({ig['speedup']} x)")
for wl, v in j.get("small_images", {}).items():
    print("small", wl, {sc: (r["frac"], round(r["avg_launch_ms"] * 1e3, 2)) for sc, r in v["scenes"].items()})
print("traffic:", r.get("traffic"), r.get("traffic_over_algorithmic"), j.get("pmc"))
s = j["sustained"]
print(f"sustained: {s['value']:.0f} Mpix/s over {s['seconds']} s, {s['gpu']}")
print("cpu_baseline:", {k: v for k, v in j["cpu_baseline"].items() if not k.endswith("note") and k != "sample"})
