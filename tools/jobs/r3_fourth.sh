#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "cloud_dev or reference_product" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|Error|error|points," $O/pytest.log | tail -20
for wl in c3_4096x3000x44 c2_1920x1080x44; do timeout 300 python tools/time_lists.py --workload $wl --knobs route=0,1 2>&1 | tail -4; done | tee $O/time_lists.log
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o lists -- python3 $OLDPWD/tools/time_lists.py --rounds 2 > /dev/null 2>&1; cd $OLDPWD
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-8 {} | head -12'
