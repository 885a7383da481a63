#!/bin/bash
# The direct-exchange tests 15 times in a row and the several-ranks-on-one-GPU suite 5 times (races in the new init / unregister / deadline code would show as a failure or a repeat)
O=gpurun_out/repeat; mkdir -p $O
pass=0; fail=0
for i in $(seq 1 15); do
  if timeout 600 python -m pytest tests/test_gpu_direct_exchange.py -q -x > $O/direct_$i.txt 2>&1; then pass=$((pass+1)); else fail=$((fail+1)); tail -30 $O/direct_$i.txt; fi
done
echo "direct exchange tests: $pass passes, $fail failures of 15" | tee $O/summary.txt
bash tools/jobs/rccl_repeat_suite.sh 5 > $O/rccl_repeat.txt 2>&1; tail -8 $O/rccl_repeat.txt | tee -a $O/summary.txt
