#!/bin/bash
# round 6, second call: tile-size sweep of the layout question; the fused compute() tests and timing; a bench run on the same box
O=gpurun_out/r6_second; mkdir -p $O
for i in 1 2; do timeout 300 tools/ubench/stream_rates L > $O/stream_rates_L_$i.txt 2>&1; done
timeout 900 python -m pytest tests/test_gpu_compute.py -x -q -s > $O/compute_tests.txt 2>&1; tail -5 $O/compute_tests.txt
timeout 600 python tools/time_dropin.py > $O/dropin.txt 2>&1; tail -12 $O/dropin.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; tail -c 1800 $O/bench.txt
cp gpurun_out/bench_extras.json $O/ 2>/dev/null
grep -h 'Dtk\|^D \|^R ' $O/stream_rates_L_1.txt
