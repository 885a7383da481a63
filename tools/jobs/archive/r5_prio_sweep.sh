#!/bin/bash
# s_setprio per phase of the fused kernel (slgc_tune "prio" = head*100 + body*10 + tail) over image sizes from one round of resident waves to many
for wl in c1_1280x720x42 c2_1920x1080x44 b8_4096x375x44 b4_4096x750x44 b2_4096x1500x44 c3_4096x3000x44; do for sc in ${SCENES:-physical s-scene}; do
  python3 tools/ab_fused.py --knobs "prio=${PRIOS:-0,200,300,210,310,201}" --workload $wl --scene $sc --rounds 5 --iters 40 2>&1 | grep -E "scene=|prio=" | cut -c1-118
done; done
