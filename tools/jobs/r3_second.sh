#!/bin/bash
# round 3, second GPU call: full -m gpu suite, occupancy-limit A/B on the 1920x1080 scan (physical scene is what bench times now)
O=gpurun_out/r3b; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -45 $O/pytest.log
timeout 300 python tools/ab_fused.py --workload c2_1920x1080x44 --rounds 4 --iters 40 --knobs "lds_pad=0,1024,2048,4096,8192,12288,24576" > $O/ab_ldspad_c2.log 2>&1
tail -12 $O/ab_ldspad_c2.log
