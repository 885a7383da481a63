# camera node table: parity at full size, A/B against the per-pixel table.   gpurun -- 'bash tools/jobs/nodes_ab.sh <tag>'
set -u
out=gpurun_out/${1:-nodes}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -3
for p in fused split; do
  timeout 600 python3 tools/ab_fused.py --knobs "cam_nodes=0,1" --pipeline $p --rounds 6 2>&1 | grep -E "cam_nodes|ray tables|DIFFER" | tee -a $out/ab.log
done
timeout 600 python3 tools/ab_fused.py --knobs "cam_nodes=0,1" --workload c2_1920x1080x44 --rounds 6 2>&1 | grep -E "cam_nodes|ray tables|DIFFER" | tee -a $out/ab.log
timeout 600 python3 tools/ab_fused.py --knobs "cam_nodes=0,1" --workload c1_1280x720x42 --rounds 6 2>&1 | grep -E "cam_nodes|ray tables|DIFFER" | tee -a $out/ab.log
