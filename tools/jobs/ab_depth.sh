#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4k; log=gpurun_out/r4k/depth.log; : > $log
for wl in c1_1280x720x42 c2_1920x1080x44 c3_4096x3000x44; do for i in 1 2 3; do for tag in base d3 d4 d4w7 d6w6; do
  if [ $tag = base ]; then unset SLGC_LIB; else export SLGC_LIB=3dscanner-graycode_amd/lib/libslgc_$tag.so; fi
  timeout 200 python3 tools/ab_fused.py --knobs "guard_list=1" --workload $wl --scene physical --rounds 3 --iters 40 2>&1 | grep "guard_list=1" | sed "s/^/$wl $tag /" | cut -c1-120 >> $log
done; done; done
python3 - <<'PY'
import re, collections
d=collections.defaultdict(list)
for ln in open("gpurun_out/r4k/depth.log"):
    m=re.match(r"(\S+) (\S+)\s+guard_list=1:\s+kernel median\s+([\d.]+) us\s+min\s+([\d.]+)", ln)
    if m: d[(m.group(1), m.group(2))].append(float(m.group(3)))
for k in sorted(d): print(k, [round(x,2) for x in d[k]], "mean", round(sum(d[k])/len(d[k]),2))
PY
