#!/bin/bash
# every bench.py variant once (short): exit codes and the executed path of each
O=gpurun_out/variants; mkdir -p $O
i=0
while read -r args; do
  i=$((i+1))
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $args > $O/v$i.json 2> $O/v$i.err; rc=$?
  python3 - "$O/v$i.json" "$rc" "$args" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("rc", sys.argv[2], "|", sys.argv[3], "| value", j.get("value"), "| pipeline", j.get("config", {}).get("pipeline"), "| executed", j.get("config", {}).get("executed", {}).get("path"),
          "| verify", (j.get("verify") or {}).get("ok"), "| err", j.get("error"))
except Exception as e:
    print("rc", sys.argv[2], "|", sys.argv[3], "| NO JSON", e)
PY
done <<'LIST'
--extras none
--pipeline split --no-throughput-mode
--mode exact --extras none
--tri direct --no-extras --workload c2_1920x1080x44
--workload b8_4096x375x44 --image-rows 3000 --extras none
--workload b8_4096x375x44 --extras none
--scene s-scene --extras none
--force-sharded --scene s-scene --extras none
--force-sharded --exchange xyz --workload c2_1920x1080x44 --extras none
--force-sharded --exchange records --no-extras --workload t_516x1031x44
--workload c3_4096x3000x46 --extras none
--workload t_516x1031x44 --extras none
--plane-pad 64 --extras none
LIST
