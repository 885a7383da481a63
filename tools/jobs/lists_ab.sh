# list-stage check: parity tests, list-stage timing, per-kernel split (rocprofv3), diagnostic ablations.   gpurun -- 'bash tools/jobs/lists_ab.sh <tag>'
set -u
out=gpurun_out/${1:-lists}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -3
for w in c3_4096x3000x44 c2_1920x1080x44 c1_1280x720x42; do
  timeout 300 python3 tools/time_lists.py --workload $w --knobs xcd=1 2>&1 | grep -E 'list|DIFFER' | tee -a $out/lists.log
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/time_lists.py --rounds 1 --knobs xcd=1 > $out/prof.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$out/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if 'xmajor' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'])
PY
export SLGC_LIB=3dscanner-graycode_amd/lib/libslgc_diag.so
for a in 0 1 2 4 7 8 16 31; do
  echo "abl $a: $(SLGC_LISTS_ABL=$a timeout 300 python3 tools/time_lists.py --knobs xcd=1 --rounds 3 2>&1 | grep 'list stage')" | tee -a $out/abl.log
done
unset SLGC_LIB
python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $out/bench.json
python3 -c "
import json;j=json.load(open('$out/bench.json'));print(j['value'],j['ms_per_step'],j['roofline']['frac'],j['roofline']['traffic']);print(j['reference_product'])"
