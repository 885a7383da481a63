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

from conftest import ROOT, has_gpu

pytestmark = pytest.mark.gpu


def n_devices():
    try:
        from scanner import _native
        return _native.device_count()
    except Exception:  # noqa: BLE001
        return 0


def run_bench(*args, timeout=600):
    env = dict(os.environ, SLGC_BENCH_TIMEOUT_S=str(timeout - 30))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


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


@pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")
@pytest.mark.parametrize("extra", [["--exchange", "maps"], ["--exchange", "maps", "--wire", "hv24"], ["--exchange", "xyz"]], ids=["maps-int16", "maps-hv24", "xyz"])
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
