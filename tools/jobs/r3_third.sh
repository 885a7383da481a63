#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_physical.py -m gpu -q -s --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|Error|error|worst|lit |points," $O/pytest.log | tail -40
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-throughput-mode > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 400 $O/bench_c3.err
timeout 300 python bench.py --workload c2_1920x1080x44 --steps 100 --warmup 10 --no-cpu-baseline --no-throughput-mode > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 400 $O/bench_c2.err
timeout 300 python bench.py --workload c1_1280x720x42 --steps 100 --warmup 10 --no-cpu-baseline --no-throughput-mode > $O/bench_c1.json 2> $O/bench_c1.err; tail -c 400 $O/bench_c1.err
