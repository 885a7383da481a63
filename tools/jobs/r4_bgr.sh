#!/bin/bash
out=gpurun_out/r4c; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_bgr_scan.py -x -q -s > $out/bgr_test.log 2>&1; echo "bgr test rc=$?" | tee -a $out/rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 --pmc off --no-small-images --no-throughput-mode --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/rc.txt
tail -15 $out/bgr_test.log; tail -3 $out/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c/bench.json"))
print(json.dumps(d.get("ingest"), indent=1))
PY
