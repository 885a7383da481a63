#!/bin/bash
# round 3, first GPU call: new tests, stagger A/B on the 1920x1080 scan, bench with the scene legs
O=gpurun_out/r3a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -x -k "not test_threshold_folding_exhaustive and not test_packed_classification_exhaustive" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
timeout 300 python tools/ab_fused.py --workload c2_1920x1080x44 --rounds 4 --iters 40 --knobs "stagger=0,552,1064,1576,1056,1059,1060,1063,1065,584,583,65576,65568,65571" > $O/ab_stagger_c2.log 2>&1
tail -20 $O/ab_stagger_c2.log
timeout 300 python bench.py --workload c2_1920x1080x44 --steps 100 --warmup 10 --no-cpu-baseline --no-throughput-mode > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.err
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-throughput-mode > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.err
