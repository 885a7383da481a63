#!/bin/bash
# where does the S-uniform scan's extra time go?  diagnostic build: generic fused kernel (park=0) with the guard / the gathers ablated
out=gpurun_out/r4b; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
for wl in c3_4096x3000x44 c2_1920x1080x44; do
  for sc in s-uniform s-scene physical; do
    SLGC_LIB=3dscanner-graycode_amd/lib/libslgc_diag.so timeout 300 python tools/ab_fused.py --knobs "park=0;guard_list=0;fuse_abl=0,8,6" --workload $wl --scene $sc --rounds 4 --iters 30 >> $out/abl.log 2>&1
    SLGC_LIB=3dscanner-graycode_amd/lib/libslgc_diag.so timeout 300 python tools/ab_fused.py --knobs "park=0;guard_list=1;fuse_abl=0" --workload $wl --scene $sc --rounds 4 --iters 30 >> $out/abl.log 2>&1
  done
done
grep -E "scene=|fuse_abl" $out/abl.log | cut -c1-180
