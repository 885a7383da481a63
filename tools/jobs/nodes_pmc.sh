# traffic of the scan kernels with the camera rays per pixel / from the node table.   gpurun -- 'bash tools/jobs/nodes_pmc.sh <tag>'
set -u
out=gpurun_out/${1:-nodes_pmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -3
for k in 0 1; do
  for p in fused split; do
    echo "== cam_nodes=$k pipeline=$p"; bash tools/pmc_knob.sh $out/pmc_${p}_$k "cam_nodes=$k" --pipeline $p 2>&1 | tail -12
  done
done | tee $out/pmc.log
