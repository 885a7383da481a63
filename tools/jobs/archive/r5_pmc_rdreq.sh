#!/bin/bash
# Cross-check of the factor 2 on FETCH_SIZE: the raw request counters behind it.  FETCH_SIZE = (BUBBLE*128 + (RDREQ - BUBBLE - RDREQ_32B)*64
# + RDREQ_32B*32) / 1024: if gfx950 tallies no "bubble" (128-byte) requests, every 128-byte request of a coalesced stream counts as 64.
# The decode kernel cannot read less than its N bytes per pixel: its request count calibrates the bytes per request; the fused kernel's gathers are
# then priced per REQUEST.  usage: tools/jobs/r5_pmc_rdreq.sh <outdir>
out=${1:-gpurun_out/r5rdreq}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
common="--steps 6 --warmup 2 --preheat 0 --extras none --pmc off --extras-file /dev/null"
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '+')
  for sc in physical s-scene s-uniform; do
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d "$out/$tag/$sc" -- python3 bench.py $common --scene $sc > /dev/null 2>&1
  done
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d "$out/$tag/decode" -- python3 bench.py $common --pipeline split --scene s-scene > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    run = f.split(os.sep)[-4] if "$out".count(os.sep) >= 0 else "?"
    scene = os.path.relpath(f, "$out").split(os.sep)[1]
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if ("k_decode_pk" in k or "k_triangulate_maps" in k) and int(row["Grid_Size"]) == 3072000:
            short = "fused" if ", 3, 44" in k else "decode" if "k_decode_pk" in k else "dense-tri"
            acc[(scene, short, row["Counter_Name"])].append(float(row["Counter_Value"]))
tab = collections.defaultdict(dict)
for (scene, kern, ctr), v in acc.items():
    tab[(scene, kern)][ctr] = sum(v) / len(v)
for (scene, kern), d in sorted(tab.items()):
    rd, r32, bub = d.get("TCC_EA0_RDREQ_sum"), d.get("TCC_EA0_RDREQ_32B_sum"), d.get("TCC_BUBBLE_sum")
    line = f"{scene:10s} {kern:9s} " + " ".join(f"{k}={v:.0f}" for k, v in sorted(d.items()))
    if rd and "FETCH_SIZE" in d:
        line += f" | FETCH_SIZE*1024/RDREQ = {d['FETCH_SIZE'] * 1024 / rd:.1f} B per request"
    print(line)
PY
