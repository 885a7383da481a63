#!/bin/bash
# The driver's command N times on one box: `value` (from the region's wall clock) against the kernel's own median -- how often does a host-side stall inside a 2.4 ms region show?
n=${1:-10}
for i in $(seq 1 $n); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras-file /tmp/x.json 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('run $i: value %.1f  ms/step %.4f  median launch %.4f  step/median %.3f  frac %.4f' % (j['value'], j['ms_per_step'], r['median_launch_ms'], j['ms_per_step'] / r['median_launch_ms'], r['frac']))"
done
