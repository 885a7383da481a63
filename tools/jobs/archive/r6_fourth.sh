#!/bin/bash
# round 6, fourth call: the default GPU suite (timed), the layout question closed on ONE box (movers and real kernels side by side), the extended suite
O=gpurun_out/r6_fourth; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu --durations=30 ) > $O/gpu_suite_default.txt 2>&1
tail -45 $O/gpu_suite_default.txt
timeout 300 tools/ubench/stream_rates L > $O/stream_rates_L.txt 2>&1
timeout 600 python tools/time_tiled.py --workload c3_4096x3000x44 --k 10 12 13 > $O/tiled_c3.txt 2>&1; cat $O/tiled_c3.txt
timeout 600 python tools/time_tiled.py --workload c3_4096x3000x44 --scene physical --k 12 > $O/tiled_c3_physical.txt 2>&1; cat $O/tiled_c3_physical.txt
grep -h '4096x3000' $O/stream_rates_L.txt | cut -c1-140
bash tools/jobs/gpu_extended.sh
