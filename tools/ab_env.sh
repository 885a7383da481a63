#!/bin/bash
# A/B on one box: tools/ab_env.sh VAR "v1 v2 ..." [bench args]   e.g.  tools/ab_env.sh SLGC_XCD "0 1"
# or plane padding:      tools/ab_env.sh --plane-pad "0 256 4352"
var=$1; vals=$2; shift 2
mkdir -p gpurun_out/ab
for i in 1 2 3; do for x in $vals; do
  if [[ $var == --* ]]; then python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-throughput-mode $var $x "$@" 2>/dev/null | tail -1 > gpurun_out/ab/b_${x}_$i.json
  else env $var=$x python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-throughput-mode "$@" 2>/dev/null | tail -1 > gpurun_out/ab/b_${x}_$i.json; fi
  python3 - <<PY
import json
j = json.load(open("gpurun_out/ab/b_${x}_$i.json"))
print("$var=$x run $i value", j["value"], "fused us", round(j["roofline"]["avg_launch_ms"] * 1e3, 2), "| split", j["split_pipeline"]["value"], "ms/step", j["split_pipeline"]["ms_per_step"],
      "decode us", round(j["split_pipeline"]["roofline"]["avg_launch_ms"] * 1e3, 2), "| decode alone us", round(j["decode_kernel_alone"]["roofline"]["avg_launch_ms"] * 1e3, 2))
PY
done; done
