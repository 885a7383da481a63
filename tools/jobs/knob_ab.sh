# one slgc_tune knob: parity, interleaved timing, HBM-side traffic.   gpurun -- 'bash tools/jobs/knob_ab.sh <tag> <knob> [values]'
set -u
out=gpurun_out/${1:-knob}; mkdir -p $out; knob=$2; vals=${3:-0,1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c3_4096x3000x44 c2_1920x1080x44; do
  timeout 600 python3 tools/ab_fused.py --knobs "$knob=$vals" --workload $w --rounds 6 2>&1 | grep -E "$knob=|DIFFER" | tee -a $out/ab.log
done
for v in ${vals//,/ }; do
  echo "== $knob=$v"; bash tools/pmc_knob.sh $out/pmc_$v "$knob=$v" 2>&1 | grep -E "k_decode|FETCH|WRITE|TCC"
done | tee $out/pmc.log
