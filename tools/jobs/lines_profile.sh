cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lines
LISTS_ROUTE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lines/kt -- python3 tools/time_lists.py --rounds 2 --knobs lists_lines=1 > gpurun_out/lines/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
LISTS_ROUTE=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/lines/pmc_$c -- python3 tools/time_lists.py --rounds 1 --iters 10 --knobs lists_lines=1 > gpurun_out/lines/pmc_$c.log 2>&1
done
timeout 1400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
