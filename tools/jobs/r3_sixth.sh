#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
for wl in c3_4096x3000x44 c2_1920x1080x44 c1_1280x720x42; do LISTS_ROUTE=1 timeout 300 python tools/time_lists.py --workload $wl --knobs lists_order=0,1,2 2>&1 | tail -4; done | tee $O/time_lists_order.log
LISTS_ROUTE=0 timeout 300 python tools/time_lists.py --knobs lists_order=0,2 2>&1 | tail -3 | tee -a $O/time_lists_order.log
