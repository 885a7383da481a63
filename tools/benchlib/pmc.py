"""bench.py: HBM traffic of the timed kernel from rocprofv3's PMC counters, collected by the run itself.

The parent process -- before it has touched the GPU -- starts one fresh child per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass:
MI355X_MICROARCH.md, rocprofv3 PMC slots): `rocprofv3 --pmc <counter> -- python3 bench.py --no-extras --steps 5 ...`, the program directly
behind `--`, no tracing flag beside --pmc.  The child's counter CSV is reduced to bytes per launch of the scan kernel:
    HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of a coalesced streaming read at 64 bytes -- the
guide's correction, calibrated here on the decode kernel, which cannot read less than its N bytes per pixel: 0.98 x algorithmic).
Round 5 checked the raw request counters behind it (tools/jobs/r5_pmc_rdreq.sh, profiles/r05_pmc_rdreq.txt): on gfx950 TCC_BUBBLE and TCC_EA0_RDREQ_32B are 0
for these kernels, so FETCH_SIZE is 64 B x TCC_EA0_RDREQ while every request is a whole 128-byte line (decode kernel: 131 B of frames per request) -- the
factor 2 holds for the streaming loads and for the table gathers alike; WRITE_SIZE is exact (64 B x TCC_EA0_WRREQ_64B)."""
import csv
import glob
import os
import shutil
import signal
import subprocess
import sys
import tempfile


def _run_child(counter, bench_py, child_args, outdir, timeout_s):
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", outdir, "--", sys.executable, bench_py] + child_args
    env = dict(os.environ, TMPDIR="/tmp", SLGC_BENCH_PMC_CHILD="1")
    try:
        p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
    except OSError as e:
        return None, f"could not start rocprofv3: {e}"
    try:
        _, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)          # the process group this call started, nothing else
        except OSError:
            pass
        p.wait()
        return None, f"rocprofv3 --pmc {counter} did not finish in {timeout_s:.0f} s"
    if p.returncode != 0:
        return None, f"rocprofv3 --pmc {counter} exited with {p.returncode}: {err.decode(errors='replace')[-300:]}"
    vals = {}
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter or "k_decode_pk" not in row.get("Kernel_Name", ""):
                continue
            vals.setdefault(int(row["Grid_Size"]), []).append(float(row["Counter_Value"]))
    if not vals:
        return None, f"no k_decode_pk dispatch in the {counter} pass"
    return vals, None


def collect(bench_py, workload, scene, pipeline, grid_size, timeout_s=90.0):
    """-> ({"fetch_size_kb", "write_size_kb", "hbm_bytes_per_launch", "launches", "source"} or None, note)"""
    if os.environ.get("SLGC_BENCH_PMC_CHILD") == "1":
        return None, "this is a counter child"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler"
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    child_args = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat", "0", "--no-extras", "--pmc", "off", "--workload", workload,
                  "--scene", scene, "--pipeline", pipeline, "--extras-file", os.devnull]
    got = {}
    top = tempfile.mkdtemp(prefix="slgc_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            vals, note = _run_child(counter, bench_py, child_args, os.path.join(top, counter), timeout_s)
            if vals is None:
                return None, note
            rows = vals.get(grid_size)
            if not rows:
                return None, f"{counter}: no dispatch with grid {grid_size} (saw {sorted(vals)})"
            got[counter] = (sum(rows) / len(rows), len(rows))
    finally:
        shutil.rmtree(top, ignore_errors=True)
    fetch_kb, write_kb = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    return {"fetch_size_kb": round(fetch_kb, 1), "write_size_kb": round(write_kb, 1),
            "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024.0)), "launches": min(got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]),
            "source": "this run: two rocprofv3 --pmc children (FETCH_SIZE, WRITE_SIZE; separate passes, no tracing) started before the parent touched the "
                      "GPU; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)"}, None
