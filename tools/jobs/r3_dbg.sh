#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
for c in plain nodes0 prior decode_only nocol nopts; do echo "== $c"; timeout 120 python tools/dbg/cloud_repro.py $c 2>&1 | grep -v "^  File\|^Extension\|^Thread\|coredump\|core dump" | tail -6; done > $O/dbg.log 2>&1
cat $O/dbg.log
