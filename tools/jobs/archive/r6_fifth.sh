#!/bin/bash
# round 6, fifth call: the fixed edge tests; the driver's launch line at N = 2, 4, 8 on one GPU (whole N > 1 path incl. the direct children); the default bench
O=gpurun_out/r6_fifth; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edges.py -q > $O/edges.txt 2>&1; tail -5 $O/edges.txt
bash tools/jobs/driver_like.sh > $O/driver_like.txt 2>&1; cat $O/driver_like.txt | cut -c1-700
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; tail -c 2600 $O/bench.txt; tail -3 $O/bench.err
cp gpurun_out/bench_extras.json $O/ 2>/dev/null
