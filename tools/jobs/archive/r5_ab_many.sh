#!/bin/bash
# Several variant libraries against the working one, processes interleaved round-robin: tools/jobs/r5_ab_many.sh <tag> name1 name2 ...   (lib/libslgc_<name>.so)
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out; : > $out/ab.log
SCENES=${SCENES:-"physical s-scene"}
WLS=${WLS:-"c1_1280x720x42 c2_1920x1080x44 c3_4096x3000x44"}
for wl in $WLS; do for sc in $SCENES; do for i in 1 2 3; do for t in base "$@"; do
  if [ $t = base ]; then unset SLGC_LIB; else export SLGC_LIB=$PWD/3dscanner-graycode_amd/lib/libslgc_$t.so; fi
  timeout 200 python3 tools/ab_fused.py --knobs "guard_list=1" --workload $wl --scene $sc --rounds 3 --iters 40 2>&1 | grep "guard_list=1" | sed "s/^/$wl $sc $t /" | cut -c1-160 >> $out/ab.log
done; done; done; done
unset SLGC_LIB
python3 - <<PY
import re, collections, statistics
rows = collections.defaultdict(list)
for ln in open("$out/ab.log"):
    p = ln.split()
    m = re.search(r"median\s+([0-9.]+)", ln)
    if m:
        rows[(p[0], p[1], p[2])].append(float(m.group(1)))
base = {}
for (wl, sc, t), v in sorted(rows.items()):
    if t == "base":
        base[(wl, sc)] = statistics.median(v)
for (wl, sc, t), v in sorted(rows.items()):
    med = statistics.median(v)
    print(f"{wl} {sc:9s} {t:10s} {med:8.2f} us  {100.0 * (med / base[(wl, sc)] - 1.0):+5.1f} %  {v}")
PY
