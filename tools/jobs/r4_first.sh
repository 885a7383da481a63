#!/bin/bash
# round 4, first GPU call: the new scene tests, guard_list A/B on every scene at C3 / C2 / C1, a default bench run
out=gpurun_out/r4a; mkdir -p $out
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_scenes.py -x -q -s > $out/scenes_test.log 2>&1; echo "scenes test rc=$?" | tee -a $out/rc.txt
for wl in c3_4096x3000x44 c2_1920x1080x44 c1_1280x720x42; do
  for sc in physical s-scene s-uniform noisy-physical; do
    timeout 300 python tools/ab_fused.py --knobs "guard_list=0,1" --workload $wl --scene $sc --rounds 4 --iters 30 >> $out/ab_guard.log 2>&1; echo "ab $wl $sc rc=$?" >> $out/rc.txt
  done
done
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?" | tee -a $out/rc.txt
tail -3 $out/scenes_test.log; grep -E "scene=|guard_list" $out/ab_guard.log | cut -c1-200
