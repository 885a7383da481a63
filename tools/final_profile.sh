#!/bin/bash
# One GPU-box call that produces everything tools/collect_profiles.py turns into profiles/rNN_*:
#   gpurun -- 'bash tools/final_profile.sh'   then (locally)   python3 tools/collect_profiles.py rNN
# (delete the local gpurun_out/final first: gpurun merges new files into it)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/final
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
# (--sustained 1 on the --extras full / traced runs: the default 5.5 s leg is 45 000 launches -- 240 000 at 1920x1080 -- and the kernel traces must fit gpurun's 64 MB)
# bench.py prints ONE compact line (kept as line_*.json) and writes the full report to --extras-file (kept as bench_*.json)
python3 bench.py --extras full --sustained 1 --extras-file "$out/bench_default.json" > "$out/bench_default.log" 2>&1
grep '^{' "$out/bench_default.log" | tail -1 > "$out/line_default.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --extras full --sustained 1 --extras-file "$out/bench_under_rocprof.json" > "$out/bench_under_rocprof.log" 2>&1
bash tools/pmc.sh "$out/pmc" > "$out/pmc.log" 2>&1
tail -3 "$out/pmc.log"
# the other BASELINE configurations (physical scene = the default; the S-scene rides along as other_scene), 1920x1080 also under the kernel trace
for w in c1_1280x720x42 c2_1920x1080x44 c3_4096x3000x46; do
  python3 bench.py --workload $w --extras full --sustained 1 --no-cpu-baseline --no-throughput-mode --extras-file "$out/bench_$w.json" 2>/dev/null | tail -1 > "$out/line_$w.json"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_c2" -- python3 bench.py --workload c2_1920x1080x44 --extras full --sustained 1 --no-cpu-baseline --no-throughput-mode --extras-file "$out/bench_c2_under_rocprof.json" > "$out/bench_c2_under_rocprof.log" 2>&1
python3 bench.py --scene s-scene --extras full --sustained 1 --no-cpu-baseline --no-throughput-mode --no-small-images --extras-file "$out/bench_sscene_headline.json" 2>/dev/null | grep '^{' | tail -1 > "$out/line_sscene_headline.json"
python3 bench.py --scene physical-survey --extras full --sustained 1 --no-cpu-baseline --no-throughput-mode --no-small-images --extras-file "$out/bench_physical_survey_headline.json" 2>/dev/null | grep '^{' | tail -1 > "$out/line_physical_survey_headline.json"
# what the driver runs, five times in a row (wall seconds, line size; VERDICT r4: no leg whose mean and median differ by > 5 %)
for i in 1 2 3 4 5; do
  t0=$(date +%s%N)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras-file "$out/bench_steps20_run$i.json" 2>/dev/null | grep '^{' | tail -1 > "$out/line_steps20_run$i.json"
  echo "driver run $i: wall $(( ($(date +%s%N) - t0) / 1000000 )) ms, line $(wc -c < "$out/line_steps20_run$i.json") bytes" >> "$out/driver_runs.txt"
done
for ex in maps xyz records; do
  python3 bench.py --force-sharded --exchange $ex --no-extras --extras-file "$out/bench_sharded_$ex.json" 2>/dev/null | grep '^{' | tail -1 > "$out/line_sharded_$ex.json"
done
python3 bench.py --force-sharded --exchange maps --exchange-impl direct --no-extras --extras-file "$out/bench_sharded_maps_direct.json" 2>/dev/null | grep '^{' | tail -1 > "$out/line_sharded_maps_direct.json"
# the "next" rows, the list stage and the whole reference-shaped product under the kernel trace, the store-pattern microbenchmark
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_next" -- python3 tools/time_next_rows.py > "$out/next_rows.log" 2>&1
LISTS_ROUTE=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_lists" -- python3 tools/time_lists.py --rounds 2 --knobs lists_lines=1 > "$out/lists.log" 2>&1
python3 tools/time_lists.py --knobs route=0,1 2>/dev/null | grep "list stage" > "$out/lists_plain.log"
LISTS_ROUTE=1 python3 tools/time_lists.py --knobs lists_lines=0,1 2>/dev/null | grep "list stage" >> "$out/lists_plain.log"
bash tools/jobs/pmc_lists.sh "$out/pmc_lists" > "$out/pmc_lists.log" 2>&1
for w in c3_4096x3000x44 c2_1920x1080x44 c1_1280x720x42; do python3 tools/time_cloud.py --workload $w 2>/dev/null | grep "per scan"; done > "$out/cloud.log"
for w in c3_4096x3000x44 c2_1920x1080x44 c1_1280x720x42; do python3 tools/time_ingest.py --workload $w 2>/dev/null | grep "scan from BGR"; done > "$out/ingest.log"
python3 tools/time_batch_rounds.py 2>/dev/null | grep "batch of" > "$out/batch_rounds.log"
for w in c3_4096x3000x44 c2_1920x1080x44; do python3 tools/time_tiled.py --workload $w --k 10 12 13 2>/dev/null; done > "$out/tiled_layout.log"     # the layout question (round 6): planar vs tile-interleaved stacks on the real kernels
for w in c3_4096x3000x44 c2_1920x1080x44 c1_1280x720x42; do for sc in physical s-scene s-uniform noisy-physical; do
  python3 tools/ab_fused.py --knobs "guard_list=0,1" --workload $w --scene $sc --rounds 4 --iters 30 2>/dev/null | grep -E "scene=|guard_list="; done; done > "$out/ab_guard.log"
python3 tools/time_host_api.py 2>/dev/null | grep Mpix > "$out/host_api.log"
python3 tools/time_dropin.py 2>/dev/null | grep -E "^pass|^c3|^  " > "$out/dropin.log"
[ -x tools/ubench/write_patterns ] && tools/ubench/write_patterns > "$out/write_patterns.txt" 2>&1
[ -x tools/ubench/stream_rates ] && tools/ubench/stream_rates > "$out/stream_rates.txt" 2>&1
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        j = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    sp, o = j.get("split_pipeline", {}), j.get("other_scene", {})
    print(os.path.basename(f), "value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "| other scene", o.get("value"), o.get("roofline", {}).get("frac"),
          "| split", sp.get("value"), sp.get("roofline", {}).get("frac"), "| alone", j.get("decode_kernel_alone", {}).get("roofline", {}).get("frac"),
          "| product", j.get("reference_product", {}).get("value"), "| thr", j.get("throughput_mode", {}).get("value"))
PY
