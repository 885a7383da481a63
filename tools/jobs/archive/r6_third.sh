#!/bin/bash
# round 6, third call: the new GPU tests (fused compute(), tiled stack layout, direct-exchange fixes), the layout A/B on the real kernels, drop-in timing, a bench run
O=gpurun_out/r6_third; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_compute.py tests/test_gpu_tiled_stack.py tests/test_gpu_direct_exchange.py -q -s > $O/new_tests.txt 2>&1; tail -15 $O/new_tests.txt
timeout 600 python tools/time_tiled.py --workload c3_4096x3000x44 > $O/tiled_c3.txt 2>&1; cat $O/tiled_c3.txt
timeout 600 python tools/time_tiled.py --workload c3_4096x3000x44 --scene s-uniform --k 12 > $O/tiled_c3_uniform.txt 2>&1; cat $O/tiled_c3_uniform.txt
timeout 600 python tools/time_tiled.py --workload c2_1920x1080x44 --k 12 > $O/tiled_c2.txt 2>&1; cat $O/tiled_c2.txt
timeout 600 python tools/time_dropin.py > $O/dropin.txt 2>&1; tail -12 $O/dropin.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; tail -c 2500 $O/bench.txt
cp gpurun_out/bench_extras.json $O/ 2>/dev/null
