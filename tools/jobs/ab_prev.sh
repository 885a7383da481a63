# new library against lib/libslgc_prev.so: parity suite (incl. the exhaustive self-tests), interleaved fused / decode timings.   gpurun -- 'bash tools/jobs/ab_prev.sh <tag>'
set -u
out=gpurun_out/${1:-abprev}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2 3; do for tag in new prev; do
  if [ $tag = prev ]; then export SLGC_LIB=3dscanner-graycode_amd/lib/libslgc_prev.so; else unset SLGC_LIB; fi
  for p in fused decode; do
    echo "$tag $i $p: $(timeout 300 python3 tools/ab_fused.py --knobs xcd=1 --pipeline $p --rounds 4 2>&1 | grep 'xcd=1')"
  done
done; done | tee $out/ab.log
