#!/bin/bash
# LDS counters of the fused scan kernel (bank conflicts of the wave-local exchange and of the parked threshold frames)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r5lds
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_ACTIVE_INST_LDS\|SQ_INST_CYCLES_[A-Z_]*\|SQ_WAIT_INST_LDS" | sort -u | tr '\n' ' '; echo
for ctrs in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
  n=$(echo $ctrs | cut -c1-24 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/$n" -- python3 bench.py --steps 10 --warmup 2 --preheat 0 --no-extras --pmc off > "$out/$n.log" 2>&1
  echo "$ctrs rc=$?"
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/${n}_cloud" -- python3 tools/time_cloud.py --workload c3_4096x3000x44 --iters 3 --rounds 1 > "$out/${n}_cloud.log" 2>&1
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/${n}_split" -- python3 bench.py --steps 10 --warmup 2 --preheat 0 --no-extras --pmc off --pipeline split --scene s-scene > "$out/${n}_split.log" 2>&1
done
python3 tools/pmc_summary.py "$out" > /dev/null 2>&1
python3 - <<PY
import json
d = json.load(open("$out/pmc_summary.json"))
for k, v in d.items():
    if "k_decode_pk" in k or "k_triangulate" in k or "k_xmajor" in k:
        print(k, {c: round(x["mean"]) for c, x in v.items()})
PY
