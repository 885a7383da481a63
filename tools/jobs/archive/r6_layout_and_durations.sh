#!/bin/bash
# round 6, first call: (1) the layout question of VERDICT r5 item 1 (Dc / Dt rows of stream_rates), twice; (2) the GPU suite with per-test durations
O=gpurun_out/r6_first; mkdir -p $O
for i in 1 2; do timeout 300 tools/ubench/stream_rates L > $O/stream_rates_L_$i.txt 2>&1; done
timeout 300 tools/ubench/stream_rates > $O/stream_rates_full.txt 2>&1
( time timeout 1500 python -m pytest tests -q -m gpu --durations=80 ) > $O/gpu_suite.txt 2>&1
tail -3 $O/gpu_suite.txt
cat $O/stream_rates_L_1.txt
