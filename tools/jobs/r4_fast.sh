#!/bin/bash
out=gpurun_out/r4g; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
timeout 2700 python -m pytest tests -x -q -m gpu > $out/suite.log 2>&1; echo "suite rc=$?" | tee $out/rc.txt
for wl in c3_4096x3000x44 c2_1920x1080x44 c1_1280x720x42; do
  for sc in physical s-scene s-uniform; do
    timeout 300 python tools/ab_fused.py --knobs "guard_list=1" --workload $wl --scene $sc --rounds 4 --iters 30 >> $out/ab.log 2>&1
  done
done
tail -6 $out/suite.log; grep -E "scene=|guard_list" $out/ab.log | cut -c1-170
