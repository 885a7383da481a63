#!/bin/bash
# Same-box A/B of two builds of the library on the fused kernel alone: tools/ab_lib_fused.sh <other.so> <out.log> [workloads...]   (processes interleaved: new, other, new, other, ...)
other=$1; log=$2; shift 2
for wl in "$@"; do for sc in physical s-scene; do for i in 1 2 3; do for tag in new other; do
  if [ $tag = other ]; then export SLGC_LIB=$other; else unset SLGC_LIB; fi
  timeout 200 python3 tools/ab_fused.py --knobs "guard_list=1" --workload $wl --scene $sc --rounds 3 --iters 40 2>&1 | grep "guard_list=1" | sed "s/^/$wl $sc $tag /" | cut -c1-150 >> $log
done; done; done; done
unset SLGC_LIB
