#!/bin/bash
# HIP-event kernel time printed by bench.py vs the rocprofv3 kernel trace of the same process.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/evchk; rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/bench.log 2>&1
grep '^{' $out/bench.log | tail -1 > $out/bench.json
python3 tools/kernel_stats_by_grid.py $(ls -t $out/kt/*/*kernel_trace.csv | head -1) $out/by_grid.csv | head -6
python3 - <<PY
import json
j = json.load(open("$out/bench.json"))
print("events: fused", round(j["roofline"]["avg_launch_ms"] * 1e3, 2), "us | split decode", round(j["split_pipeline"]["roofline"]["avg_launch_ms"] * 1e3, 2),
      "us | decode alone", round(j["decode_kernel_alone"]["roofline"]["avg_launch_ms"] * 1e3, 2), "us | value", j["value"])
PY
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-throughput-mode | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('plain run: value', j['value'], 'fused us', round(j['roofline']['avg_launch_ms']*1e3,2))"
