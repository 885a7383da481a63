#!/bin/bash
# The profile of the round (tools/final_profile.sh) and the soak on the final sources
bash tools/final_profile.sh > gpurun_out/final_profile.log 2>&1; tail -25 gpurun_out/final_profile.log
O=gpurun_out/soak; mkdir -p $O
( time SLGC_FUZZ_SCALE=300 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bgr_scan.py -k fuzz -q ) > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt
( time SLGC_DIRECT_TEST_ROUNDS=400 timeout 900 python -m pytest tests/test_gpu_direct_exchange.py -q ) > $O/direct_soak.txt 2>&1; tail -4 $O/direct_soak.txt
