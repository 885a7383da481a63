# kernels of the "next" rows under the kernel trace.   gpurun -- 'bash tools/jobs/next_rows.sh <tag> [lib]'
set -u
out=gpurun_out/${1:-next}; mkdir -p $out
[ -n "${2:-}" ] && export SLGC_LIB=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "ingest or diff or gray or outlier or knn or bad_images" > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/time_next_rows.py > $out/run.log 2>&1
grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl|rocprofv3|output_stream" $out/run.log | tail -8
python3 - <<PY
import csv,glob
for f in glob.glob("$out/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('bgr','frame_diff','knn','cell')): print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
