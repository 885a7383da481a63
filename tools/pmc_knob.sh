#!/bin/bash
# HBM-side traffic of the scan kernel under one slgc_tune setting: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; never combined
# with tracing) around a few launches, summarised per kernel @ grid.   usage: tools/pmc_knob.sh <outdir> "<knob=v;knob=v>" [ab_fused args]
set -u
out=$1; knobs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p "$out"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 tools/ab_fused.py --knobs "$knobs" --rounds 1 --iters 6 --preheat 0 "$@" > "$out/pass$i.log" 2>&1
  echo "pass $i ($ctrs): rc=$?"
done
python3 tools/pmc_summary.py "$out" | grep -A12 "k_decode_pk.*@grid=3072000\|k_triangulate_maps_lds.*@grid=3072000" | grep -E "k_decode|k_tri|FETCH|WRITE|VALU |TCC" 
