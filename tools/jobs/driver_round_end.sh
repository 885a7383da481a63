#!/bin/bash
# What the driver runs at round end, on the final sources -- pytest -m gpu, smoke(), bench.py
O=gpurun_out/round_end; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/gpu_suite.txt 2>&1; tail -6 $O/gpu_suite.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; tail -c 700 $O/bench.txt; tail -2 $O/bench.err
