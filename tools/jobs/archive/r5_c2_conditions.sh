#!/bin/bash
# Why does the fused kernel at 1920x1080 take 25.2 us in bench.py's own timed region and 24.1 us in the small-images leg of the same box?
# The same headline step under different conditions of the timed region.
for args in "" "--preheat 0.6" "--preheat 1.5" "--event-stride 1" "--event-stride 2" "--preheat 0" "--steps 40 --warmup 5" "--steps 40 --warmup 5 --event-stride 1" "--buffers 2" "--buffers 8"; do
  for i in 1 2; do
    python3 bench.py --workload c2_1920x1080x44 --extras none --pmc off --extras-file /dev/null $args 2>/dev/null | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('c2 [$args] run $i: median', round(r['median_launch_ms'] * 1e3, 2), 'us mean', round(r['avg_launch_ms'] * 1e3, 2), 'frac', r['frac'], 'ms/step', j['ms_per_step'], 'n', r['launches_timed'])"
  done
done
