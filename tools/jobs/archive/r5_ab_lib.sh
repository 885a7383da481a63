#!/bin/bash
# Same-box A/B of the working library against another build (default lib/libslgc_r5base.so = the round's starting point):
#   1. the fused kernel alone, processes interleaved, three sizes x SCENES (tools/ab_fused.py kernel medians)
#   2. HBM traffic of the fused kernel from the run's own counter children (bench.py --pmc on), both libraries, at 4096x3000
# usage: tools/jobs/r5_ab_lib.sh <tag> [other.so]
tag=${1:-r5ab}; other=${2:-3dscanner-graycode_amd/lib/libslgc_r5base.so}
out=gpurun_out/$tag; mkdir -p $out
SCENES=${SCENES:-"physical s-scene s-uniform"}
WLS=${WLS:-"c3_4096x3000x44 c2_1920x1080x44 c1_1280x720x42"}
: > $out/ab.log
for wl in $WLS; do for sc in $SCENES; do for i in 1 2 3; do for t in new other; do
  if [ $t = other ]; then export SLGC_LIB=$other; else unset SLGC_LIB; fi
  timeout 200 python3 tools/ab_fused.py --knobs "guard_list=1" --workload $wl --scene $sc --rounds 3 --iters 40 2>&1 | grep "guard_list=1" | sed "s/^/$wl $sc $t /" | cut -c1-160 >> $out/ab.log
done; done; done; done
unset SLGC_LIB
python3 - <<PY
import re, collections, statistics
rows = collections.defaultdict(list)
for ln in open("$out/ab.log"):
    p = ln.split()
    m = re.search(r"median\s+([0-9.]+)", ln) or re.search(r"med[a-z]*[ =]+([0-9.]+)", ln)
    if m:
        rows[(p[0], p[1], p[2])].append(float(m.group(1)))
for (wl, sc, t), v in sorted(rows.items()):
    print(wl, sc, t, "median of medians us", round(statistics.median(v), 2), v)
PY
if [ "${PMC:-1}" = 1 ]; then
for sc in $SCENES; do for t in new other; do
  if [ $t = other ]; then export SLGC_LIB=$other; else unset SLGC_LIB; fi
  python3 bench.py --steps 20 --warmup 5 --scene $sc --extras none --pmc on --extras-file $out/pmc_${sc}_$t.json 2>/dev/null | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('traffic $sc $t', r['traffic'], 'x algorithmic', r['traffic_over_algorithmic'], 'frac', r['frac'], 'median ms', r['median_launch_ms'])"
done; done
unset SLGC_LIB
fi
