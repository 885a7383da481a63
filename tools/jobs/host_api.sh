# host-buffer API: parity suite + PCIe-inclusive rates, with and without the parallel download.   gpurun -- 'bash tools/jobs/host_api.sh <tag>'
set -u
out=gpurun_out/${1:-host}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for v in 0 1; do
  echo "SLGC_PAR_DOWNLOAD=$v"
  SLGC_PAR_DOWNLOAD=$v timeout 300 python3 tools/time_pcie.py 2>&1 | grep -E "ms"
  SLGC_PAR_DOWNLOAD=$v timeout 600 python3 tools/time_host_api.py 2>&1 | grep -E "Mpix"
done | tee $out/host.log
