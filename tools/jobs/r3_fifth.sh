#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$O/prof -o lists -- python3 $root/tools/time_lists.py --rounds 2 --knobs route=1 > $root/$O/lists_prof.log 2>&1
cd $root
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f; cut -d, -f1-7 $f | head -12
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-throughput-mode > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.err
python3 - <<PY
import json
j=json.loads(open("$O/bench_c3.json").read().strip().splitlines()[-1])
r=j["reference_product"]; print({k:r[k] for k in ("value","ms_per_scan","list_stage_ms")}, r["list_stage_roofline"]["frac"], r["via_dense_xyz"]["ms_per_scan"])
PY
