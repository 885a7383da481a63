#!/bin/bash
# The whole GPU suite including the extended cases (tests/conftest.py: `extended`): what `pytest -m gpu` ran up to round 5.
O=gpurun_out/gpu_extended; mkdir -p $O
( time SLGC_GPU_EXTENDED=1 timeout 1700 python -m pytest tests -q -m gpu --durations=15 ) > $O/suite.txt 2>&1
tail -25 $O/suite.txt
