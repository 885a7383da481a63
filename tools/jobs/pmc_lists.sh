#!/bin/bash
# PMC counters of the x-major list build of slgc_cloud_dev (k_xmajor_lines / k_xmajor_scatter<short, 2, 64> with LINES=0): separate passes, --pmc only.
# usage: [LINES=0|1] tools/jobs/pmc_lists.sh <outdir>     -> <outdir>/summary.txt
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  LISTS_ROUTE=1 timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 tools/time_lists.py --rounds 1 --iters 10 --knobs lists_lines=${LINES:-1} > "$out/pass$i.log" 2>&1
  echo "pass $i ($ctrs): rc=$?"
done
python3 - "$out" <<'PY' | tee "$out/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "xmajor" in k:
            acc[k[k.index("k_xmajor"):].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v)
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {mean(v):16.1f}  (n={len(v)})")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        print(f"   HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB = {(2 * mean(d['FETCH_SIZE']) + mean(d['WRITE_SIZE'])) * 1024 / 1e6:.1f} MB")
    if "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d:
        print(f"   vector ALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs = {mean(d['SQ_ACTIVE_INST_VALU']) * 4 / 1024 / (mean(d['GRBM_GUI_ACTIVE']) / 8):.2f}")
    if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
        print(f"   vector instructions per wave = {mean(d['SQ_INSTS_VALU']) / mean(d['SQ_WAVES']):.0f}")
PY
