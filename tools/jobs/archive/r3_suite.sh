#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 --durations=8 -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^chain|passed|failed|FAILED|Error|pytest rc" $O/pytest.log | tail -30
