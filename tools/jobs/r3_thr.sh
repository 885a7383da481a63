#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
ab() { # workload pipeline
for i in 1 2 3; do for tag in new old; do
  if [ $tag = old ]; then export SLGC_LIB=$PWD/3dscanner-graycode_amd/lib/libslgc_oldthr.so; else unset SLGC_LIB; fi
  timeout 200 python3 tools/ab_fused.py --workload $1 --pipeline $2 --knobs park=1 --rounds 3 --iters 40 2>/dev/null | grep "park=1" | sed "s/^/$tag $1 $2 /"
done; done; unset SLGC_LIB; }
{ ab c3_4096x3000x44 fused; ab c3_4096x3000x44 decode; ab c2_1920x1080x44 fused; ab c2_1920x1080x44 decode; } | tee $O/ab_thresholds.log
