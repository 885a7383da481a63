"""RCCL with more than one rank (-m gpu; needs >= 2 HIP devices, skips on the one-GPU test box).

RCCL refuses two ranks on one device, so this is the only place where the product's exchange -- the in-place ncclAllGather path for
equal bands, the grouped per-root ncclBroadcast path for ragged bands, the communication stream and its events, the 3-byte wire
format -- runs with nranks > 1.  Every case is a self-verifying bench.py run (one process per GPU, started by bench.py itself): after
the timed, pipelined scans each rank compares the reassembled maps (bit-exact) and an XYZ sample with a single-GPU fused scan of the same
stack and its digest with every other rank's; a failed check is a non-zero exit."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, ext, extended, has_gpu

pytestmark = pytest.mark.gpu


def n_devices():
    try:
        from scanner import _native
        return _native.device_count()
    except Exception:  # noqa: BLE001
        return 0


STALL_DIR = os.path.join(ROOT, "gpurun_out", "rccl_stalls")


def keep_stall_evidence(tag, attempt, stdout, stderr):
    """A run that stopped making progress is evidence, not noise: its JSON line names the stage it was in (bench.py's watchdog) and, with
    SLGC_BENCH_FAULTHANDLER_S set, every rank has dumped its Python stacks to stderr.  Kept under gpurun_out/ (merged back from the GPU box)
    and listed in pytest's terminal summary; the test emits a warning as well."""
    import warnings

    import conftest
    os.makedirs(STALL_DIR, exist_ok=True)
    where = os.path.join(STALL_DIR, f"{tag}_attempt{attempt}.txt")
    with open(where, "w") as f:
        f.write("==== stdout (tail)\n" + (stdout or "")[-6000:] + "\n==== stderr (tail)\n" + (stderr or "")[-20000:])
    conftest.RCCL_STALLS.append((tag, attempt, os.path.relpath(where, ROOT)))
    warnings.warn(f"RCCL-on-one-GPU run {tag} attempt {attempt} timed out and was repeated; evidence in {os.path.relpath(where, ROOT)}")


def load_report(lines, side_file):
    """bench.py prints ONE compact line (the driver's contract: < 4 KB, fixed keys) and writes the full report to the side file: the tests read
    the full report, after checking the line against it."""
    if not lines:
        return None
    line = json.loads(lines[-1])
    assert len(lines[-1]) < 4096, len(lines[-1])
    if not os.path.exists(side_file):
        return line                                         # error lines (no report was assembled)
    try:
        with open(side_file) as f:
            full = json.load(f)
    finally:
        os.unlink(side_file)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype"):
        assert line[k] == full[k], (k, line[k], full[k])
    if full.get("verify") is not None:
        assert line["verify_ok"] == bool(full["verify"]["ok"])
    if full.get("sharded"):
        assert line["sharded"]["rccl_nranks"] == full["sharded"]["rccl_nranks"]
    return full


def run_bench(*args, timeout=600, ranks_as_hosts=False, attempts=1, tag="run"):
    """attempts > 1 (only the several-ranks-on-ONE-GPU test mode passes it): a run that TIMES OUT is repeated ONCE MORE AT MOST per extra attempt
    -- after its evidence has been kept (keep_stall_evidence) and reported; a run that finishes with a wrong result or an error is never
    repeated.  Round 4 looked for the stall with tools/jobs/rccl_hang_hunt.py: 170 runs of these cases (2 - 8 ranks, every strategy) on two
    boxes, none stalled (NOTES.md); the repeat stays as a fence, loud instead of silent."""
    env = dict(os.environ, SLGC_BENCH_TIMEOUT_S=str(timeout - 30))
    if ranks_as_hosts:
        env["SLGC_RANKS_AS_HOSTS"] = "1"
        env.setdefault("SLGC_BENCH_FAULTHANDLER_S", str(max(10, timeout - 40)))      # every rank dumps its Python stacks shortly before the watchdog fires
    import tempfile
    for attempt in range(attempts):
        side = tempfile.NamedTemporaryFile(prefix="slgc_bench_", suffix=".json", delete=False)
        side.close()
        os.unlink(side.name)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--extras-file", side.name], capture_output=True, text=True,
                           timeout=timeout + 60, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        j = load_report(lines, side.name)
        timed_out = j is None or "timed out" in str(j.get("error", ""))
        if not (timed_out and r.returncode != 0) or attempt + 1 == attempts:
            return r, j
        keep_stall_evidence(tag, attempt + 1, r.stdout, r.stderr)
    return r, j


@pytest.mark.skipif(not has_gpu() or n_devices() < 2, reason="needs >= 2 HIP devices (RCCL refuses two ranks on one)")
@pytest.mark.parametrize("exchange,wire", [("maps", "int16"), ("maps", "hv24"), ("xyz", "int16"), ("records", "int16")])
@pytest.mark.parametrize("workload", ["t_512x1024x44", "t_516x1031x44", "c3_4096x3000x44"])     # even bands (ncclAllGather), ragged (grouped broadcasts), full size
def test_two_ranks_self_verifying(exchange, wire, workload):
    if exchange == "records" and workload != "t_516x1031x44":
        pytest.skip("records strategy: one case")
    r, j = run_bench("--gpus", "2", "--steps", "41", "--warmup", "3", "--no-extras", "--workload", workload, "--exchange", exchange, "--wire", wire)
    assert r.returncode == 0 and j is not None, r.stderr[-3000:]
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["sharded"]["rccl_nranks"] == 2
    if exchange != "records":
        assert j["verify"]["ok"] and j["verify"]["ranks_hold_identical_results"] and j["verify"]["maps_equal_single_gpu_scan"], j["verify"]


@pytest.mark.skipif(not has_gpu() or n_devices() < 4, reason="needs >= 4 HIP devices")
def test_four_ranks_ragged_and_even():
    for workload in ("t_516x1031x44", "c3_4096x3000x44"):
        r, j = run_bench("--gpus", "4", "--steps", "20", "--warmup", "3", "--no-extras", "--workload", workload)
        assert r.returncode == 0 and j["verify"]["ok"], r.stderr[-3000:]


@pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")
def test_single_rank_rccl_through_bench_self_verifies():
    """nranks = 1 through the same path (what the one-GPU box can run): both pipelined strategies + the ragged test workload."""
    for extra in (["--exchange", "maps"], ["--exchange", "xyz"], ["--exchange", "maps", "--wire", "hv24"]):
        r, j = run_bench("--force-sharded", "--steps", "12", "--warmup", "2", "--no-extras", "--workload", "t_516x1031x44", *extra)
        assert r.returncode == 0 and j["verify"]["ok"], (extra, r.stderr[-2000:])
        assert j["sharded"]["rccl_nranks"] == 1 == j["sharded"]["distinct_devices"] and j["sharded"]["rccl_nranks_source"] == "ncclCommCount", j["sharded"]


@pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")
@pytest.mark.parametrize("extra", [ext(["--exchange", "maps"], id="maps-int16"), pytest.param(["--exchange", "maps", "--wire", "hv24"], id="maps-hv24"),
                                   pytest.param(["--exchange", "xyz"], id="xyz")])
def test_single_rank_rccl_full_size_self_verifies(extra):
    """BASELINE.json configs[3]'s image (4096x3000x44) through every RCCL call of the sharded path at nranks = 1: in-place ncclAllGather on the
    communication stream, event slots, pipelined submit/flush -- and the run's own check against a single-GPU fused scan, bit for bit."""
    r, j = run_bench("--force-sharded", "--steps", "12", "--warmup", "2", "--no-extras", "--workload", "c3_4096x3000x44", *extra)
    assert r.returncode == 0 and j is not None, r.stderr[-2000:]
    v = j["verify"]
    assert v["ok"] and v["maps_equal_single_gpu_scan"] and v["xyz_sample_equal_single_gpu_scan"] and v["valid_pixels"] > 1_000_000, v
    assert "configs[3]" in j["config"]["workload"] and j["sharded"]["rccl_nranks"] == 1


@pytest.mark.skipif(not has_gpu() or n_devices() != 1, reason="the refusal only happens on a one-GPU box")
def test_two_ranks_on_one_gpu_fail_cleanly():
    """RCCL's "Duplicate GPU detected": both rank processes must exit (no hang), rank 0 still prints a JSON line naming the error."""
    r, j = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras", "--workload", "t_512x1024x44", timeout=180)
    assert r.returncode != 0 and j is not None and j["value"] is None and "RCCL" in j["error"]


# ------------------------------------------------------------------------------------------------ RCCL with nranks > 1 on ONE GPU
# RCCL refuses two ranks on one device of one host.  With SLGC_RANKS_AS_HOSTS=1 bench.py gives every rank process a host name of its own
# (NCCL_HOSTID) and restricts RCCL to the loopback socket transport: the ranks then form a real communicator of nranks > 1 on the single GPU
# of the test box, and every RCCL call of the sharded path -- in-place ncclAllGather for equal bands, grouped ncclBroadcast for ragged ones,
# the pair exchange of the two maps in one group, the communication stream and its four event slots under the pipelined submit / flush, the
# 3-byte wire, the count all-gather + all-gatherv of the records strategy -- runs exactly as it will over xGMI, at the speed of a TCP socket.
# The run verifies itself: every rank compares what it reassembled, bit for bit, with a single-GPU fused scan of the same stack and its digest
# with every other rank's.  What this does not cover is the xGMI transport itself and its speed.
def _skip_if_transport_unavailable(r, j):
    """The loopback socket transport is a property of the box, not of the code under test: a run that dies while the communicator is being
    formed (no usable `lo`, sockets forbidden) skips; anything later -- a wrong result, a hang, an error in a collective -- fails."""
    err = str((j or {}).get("error", ""))
    if j is not None and j.get("value") is None and ("slgc_comm_init" in err or "CommInitRank" in err or "unique id" in err):
        pytest.skip(f"RCCL could not form a communicator over the loopback socket transport on this box: {err[:300]}")


EXTENDED_CASES = {1, 5, 6, 8}                                              # indices into CASES that only run with SLGC_GPU_EXTENDED=1 (same paths at other rank counts)
CASES = [
    (2, "t_512x1024x44", ["--exchange", "maps", "--wire", "int16"]),       # even bands: in-place ncclAllGather, both maps in one group
    (2, "t_512x1024x44", ["--exchange", "maps", "--wire", "hv24"]),        # packed 3 B/pixel wire, unpacked inside the triangulation kernel
    (3, "t_516x1031x44", ["--exchange", "maps", "--wire", "int16"]),       # ragged bands: grouped ncclBroadcast per contributing rank
    (2, "t_516x1031x44", ["--exchange", "xyz"]),                           # fused kernel per band, three in-place band exchanges
    (4, "t_516x1031x44", ["--exchange", "records"]),                       # count all-gather + all-gatherv of 16-byte records
    (5, "t_512x1024x44", ["--exchange", "xyz", "--no-overlap"]),           # more ranks than divide the rows evenly, one scan at a time
    # the direct exchange (csrc/direct.hip): bands pushed into the peers' buffers through hipIpcMemHandle mappings, flags in a shared segment.
    # IPC mappings work between processes on one GPU, so everything but the links themselves runs here: registration, gate / push / flag
    # kernels, the release protocol under the pipelined submit / flush, ragged bands, three buffers per exchange
    (2, "t_512x1024x44", ["--exchange", "maps", "--wire", "int16", "--exchange-impl", "direct"]),
    (3, "t_516x1031x44", ["--exchange", "maps", "--wire", "hv24", "--exchange-impl", "direct"]),
    (2, "t_516x1031x44", ["--exchange", "xyz", "--exchange-impl", "direct"]),
    (5, "t_512x1024x44", ["--exchange", "xyz", "--no-overlap", "--exchange-impl", "direct"]),
]


@pytest.mark.rccl_one_gpu
@pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")
@pytest.mark.parametrize("nranks,workload,extra", [pytest.param(n, w, e, id=f"{n}x-{w.split('_')[1]}-{'-'.join(e).replace('--', '')}",
                                                                marks=[extended] if i in EXTENDED_CASES else []) for i, (n, w, e) in enumerate(CASES)])
def test_rccl_several_ranks_on_one_gpu_over_loopback(nranks, workload, extra):
    r, j = run_bench("--gpus", str(nranks), "--steps", "9", "--warmup", "2", "--no-extras", "--scene", "s-scene", "--workload", workload, *extra,
                     timeout=75, ranks_as_hosts=True, attempts=2, tag=f"{nranks}x-{workload}-{'-'.join(extra).replace('--', '')}")      # a healthy run takes 3-10 s
    _skip_if_transport_unavailable(r, j)
    assert r.returncode == 0 and j is not None and j.get("value"), (r.stdout[-1500:], r.stderr[-3000:])
    assert j["n_gpus"] == nranks and j["sharded"]["rccl_nranks"] == nranks and j["valid_pixels_per_scan"] > 1000
    # RCCL's own account of the job: the rank count is ncclCommCount's, every rank's GPU is named -- here all the same one (test mode)
    sh = j["sharded"]
    assert sh["rccl_nranks_source"] == "ncclCommCount" and sh["rccl_info_error"] is None and sh["rccl_rank"] == 0
    assert len(sh["rank_devices"]) == nranks and len(set(sh["rank_devices"])) == 1 == sh["distinct_devices"] and sh["ranks_share_gpus_test_mode"] is True
    assert ":" in sh["rank_devices"][0]
    if "records" not in extra:
        v = j["verify"]
        assert v["ok"] and v["ranks_hold_identical_results"] and v["maps_equal_single_gpu_scan"] and v["xyz_sample_equal_single_gpu_scan"], v
        assert v["valid_pixels"] > 1000


@pytest.mark.rccl_one_gpu
@pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")
@pytest.mark.parametrize("nranks,extra", [ext(2, ["--exchange", "maps"], id="2-maps"), pytest.param(8, ["--exchange", "maps"], id="8-maps"),
                                          ext(7, ["--exchange", "xyz"], id="7-xyz-ragged"),
                                          pytest.param(8, ["--exchange", "maps", "--exchange-impl", "direct"], id="8-maps-direct"),      # (55 s with acquire-load polls, 6 s since the polls are relaxed loads: round 6)
                                          pytest.param(7, ["--exchange", "xyz", "--exchange-impl", "direct"], id="7-xyz-ragged-direct")])
def test_rccl_configs3_full_size_on_one_gpu_over_loopback(nranks, extra):
    """BASELINE.json configs[3] -- 4096x3000x44 row-sharded over 2 / 8 / (ragged) 7 ranks -- through the real RCCL exchange (loopback socket
    transport, all ranks on the one GPU), pipelined, self-verified bit for bit on every rank."""
    r, j = run_bench("--gpus", str(nranks), "--steps", "5", "--warmup", "1", "--no-extras", "--workload", "c3_4096x3000x44", *extra,
                     timeout=90, ranks_as_hosts=True, attempts=2, tag=f"configs3-{nranks}x-{'-'.join(extra).replace('--', '')}")       # a healthy run takes 5-10 s
    _skip_if_transport_unavailable(r, j)
    assert r.returncode == 0 and j is not None and j.get("value"), (r.stdout[-1500:], r.stderr[-3000:])
    v = j["verify"]
    assert j["sharded"]["rccl_nranks"] == nranks and "configs[3]" in j["config"]["workload"]
    assert j["sharded"]["distinct_devices"] == 1 and len(j["sharded"]["rank_devices"]) == nranks
    assert v["ok"] and v["ranks_hold_identical_results"] and v["maps_equal_single_gpu_scan"] and v["xyz_sample_equal_single_gpu_scan"] and v["valid_pixels"] > 1_000_000, v


@pytest.mark.rccl_one_gpu
@pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")
def test_driver_launcher_two_ranks_on_one_gpu():
    """The driver's own launch line -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`` -- with the complete
    default N > 1 run behind it (main strategy, compute-only leg, self-verification, throughput mode with its barriers, the alternative
    exchanges): RANK / LOCAL_RANK / WORLD_SIZE from the environment, the RCCL id through a file, ONE JSON line from rank 0, exit code 0."""
    pytest.importorskip("torch")
    env = dict(os.environ, SLGC_RANKS_AS_HOSTS="1", SLGC_BENCH_TIMEOUT_S="150", SLGC_BENCH_ALT_TIMEOUT_S="90", SLGC_BENCH_FAULTHANDLER_S="140")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29611",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--workload", "c2_1920x1080x44", "--no-cpu-baseline",
           "--extras-file", os.path.join(ROOT, "gpurun_out", "bench_extras_driver_launcher_test.json")]
    for attempt in range(2):                                                # a run that times out is repeated once, loudly (run_bench's docstring); a wrong result never
        env["MASTER_PORT"] = cmd[cmd.index("--master-port") + 1] = str(29611 + attempt)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        except subprocess.TimeoutExpired as e:
            keep_stall_evidence("driver-launcher-2x", attempt + 1, e.stdout if isinstance(e.stdout, str) else (e.stdout or b"").decode(errors="replace"),
                                e.stderr if isinstance(e.stderr, str) else (e.stderr or b"").decode(errors="replace"))
            if attempt == 1:
                raise
            continue
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        first = json.loads(lines[-1]) if lines else None
        if r.returncode != 0 and (first is None or "timed out" in str(first.get("error", "")) + str(first.get("sharded_alternatives", ""))) and attempt < 1:
            keep_stall_evidence("driver-launcher-2x", attempt + 1, r.stdout, r.stderr)
            continue
        break
    _skip_if_transport_unavailable(r, json.loads(lines[-1]) if lines else None)
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    j = load_report(lines, os.path.join(ROOT, "gpurun_out", "bench_extras_driver_launcher_test.json"))
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["scaling"] == "strong" and j["sharded"]["rccl_nranks"] == 2 and j["verify"]["ok"]
    alt = j["sharded_alternatives"]
    assert j["sharded"]["wire"] == "hv24"                                   # the default wire with more than one rank: 3 B/pixel
    assert alt["maps_int16"]["maps_equal_main_strategy_on_every_rank"] and alt["xyz"]["maps_equal_main_strategy_on_every_rank"], alt
    assert alt["maps_hv24_direct"]["exchange_impl"] == "direct" and alt["maps_hv24_direct"]["maps_equal_main_strategy_on_every_rank"], alt
    assert j["throughput_mode"]["value"] > 0 and j["throughput_mode"]["scaling"] == "weak"
