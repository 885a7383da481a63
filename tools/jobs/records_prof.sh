# kernels of the records strategy (fused scan -> compaction -> all-gatherv) at nranks = 1 under the kernel trace.   gpurun -- 'bash tools/jobs/records_prof.sh <tag>'
set -u
out=gpurun_out/${1:-records}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --force-sharded --exchange records --no-extras --no-cpu-baseline > $out/run.log 2>&1
grep '^{' $out/run.log | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
python3 - <<PY
import csv,glob
for f in glob.glob("$out/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:80], r['Calls'], r['AverageNs'], r['MinNs'])
PY
