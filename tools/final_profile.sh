#!/bin/bash
# One GPU-box call that produces everything tools/collect_profiles.py turns into profiles/rNN_*:
#   gpurun -- 'bash tools/final_profile.sh'   then (locally)   python3 tools/collect_profiles.py rNN
# (delete the local gpurun_out/final first: gpurun merges new files into it)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/final
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
python3 bench.py > "$out/bench_default.log" 2>&1
grep '^{' "$out/bench_default.log" | tail -1 > "$out/bench_default.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py > "$out/bench_under_rocprof.log" 2>&1
bash tools/pmc.sh "$out/pmc" > "$out/pmc.log" 2>&1
tail -3 "$out/pmc.log"
ls "$out" "$out/kt"/* | head -20
