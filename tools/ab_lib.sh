#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <other.so> [bench args]   (SLGC_LIB selects the library)
other=$1; shift
for i in 1 2 3; do for tag in new old; do
  if [ $tag = old ]; then export SLGC_LIB=$other; else unset SLGC_LIB; fi
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$tag run $i value', j['value'], 'fused us', round(j['roofline']['avg_launch_ms'] * 1e3, 2), '| split', j.get('split_pipeline', {}).get('value'), '| thr', j.get('throughput_mode', {}).get('value'))"
done; done
