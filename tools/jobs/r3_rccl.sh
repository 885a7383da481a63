#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_rccl_multi.py -m gpu -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
