#!/bin/bash
# Collect rocprofv3 PMC counters for the bench kernels (separate passes; never combined with tracing).
# usage: tools/pmc.sh <outdir> [bench args...]
# Per counter group three short bench runs, each launching ONE kind of scan kernel on ONE scene (the summary is keyed by kernel @ grid):
#   passN          fused kernel on the headline (physical, covering rig) scene   passN_split  decode kernel + dense triangulation kernel (S-scene stacks)
#   passN_sscene   fused kernel on the S-scene                                   passN_c2     (FETCH_SIZE / WRITE_SIZE only) fused kernel at 1920x1080, physical scene
#   passN_suniform (FETCH_SIZE / WRITE_SIZE only) fused kernel on S-uniform      passN_bgr    (FETCH_SIZE / WRITE_SIZE only) the fused kernel reading BGR frames (tools/time_ingest.py)
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  common="--steps 10 --warmup 2 --preheat 0 --no-extras --pmc off"
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 bench.py $common "$@" > "$out/pass$i.log" 2>&1
  rc1=$?
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass${i}_split" -- python3 bench.py $common --pipeline split --scene s-scene "$@" > "$out/pass${i}_split.log" 2>&1
  rc2=$?
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass${i}_sscene" -- python3 bench.py $common --scene s-scene "$@" > "$out/pass${i}_sscene.log" 2>&1
  rc3=$?
  if [ $i -le 2 ]; then      # HBM bytes of the fused kernel at 1920x1080 (BASELINE configs[1]), on S-uniform and reading BGR frames as well
    timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass${i}_c2" -- python3 bench.py $common --workload c2_1920x1080x44 > "$out/pass${i}_c2.log" 2>&1
    timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass${i}_suniform" -- python3 bench.py $common --scene s-uniform "$@" > "$out/pass${i}_suniform.log" 2>&1
    timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass${i}_bgr" -- python3 tools/time_ingest.py --iters 6 > "$out/pass${i}_bgr.log" 2>&1
  fi
  echo "pass $i ($ctrs): rc=$rc1 $rc2 $rc3 $?"
done
python3 tools/pmc_summary.py "$out"
