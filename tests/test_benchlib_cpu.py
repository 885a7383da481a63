"""bench.py's support code that needs no GPU: the run's own counter collection (tools/benchlib/pmc.py) against a stand-in `rocprofv3`, the scene registry,
the command line."""
import os
import stat
import subprocess
import sys

import pytest

import bench
from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib import pmc  # noqa: E402

FAKE = r'''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
counter = a[a.index("--pmc") + 1]
out = a[a.index("-d") + 1]
child = a[a.index("--") + 1:]
assert "--pmc" in child and child[child.index("--pmc") + 1] == "off" and "--no-extras" in child      # the child must not start children of its own
assert os.environ.get("SLGC_BENCH_PMC_CHILD") == "1"
mode = os.environ.get("FAKE_ROCPROF_MODE", "ok")
if mode == "fail":
    sys.exit(3)
if mode == "hang":
    import time
    time.sleep(60)
os.makedirs(os.path.join(out, "host"), exist_ok=True)
rows = ["Kernel_Name,Grid_Size,Counter_Name,Counter_Value"]
val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 300.0}[counter]
for k in range(4):
    rows.append(f'"void k_decode_pk<4, 128, 1, false, 0, 3, 44, 0>(PkArgs)",3072000,{counter},{val + k}')
rows.append(f'"void k_decode_pk<4, 128, 1, false, 0, 3, 44, 0>(PkArgs)",518400,{counter},7.0')      # another grid: ignored
rows.append(f'"k_synth(SynthArgs)",3072000,{counter},99999.0')                                      # another kernel: ignored
open(os.path.join(out, "host", "1_counter_collection.csv"), "w").write("\n".join(rows) + "\n")
'''


@pytest.fixture
def fake_rocprof(tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("SLGC_BENCH_PMC_CHILD", raising=False)
    return exe


def test_counter_children_are_folded_into_bytes_per_launch(fake_rocprof, monkeypatch):
    live, note = pmc.collect(os.path.join(ROOT, "bench.py"), "c3_4096x3000x44", "physical", "fused", grid_size=3072000, timeout_s=30)
    assert note is None and live["launches"] == 4
    assert live["fetch_size_kb"] == 1001.5 and live["write_size_kb"] == 301.5
    assert live["hbm_bytes_per_launch"] == round((2 * 1001.5 + 301.5) * 1024)           # gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes
    # no dispatch of the timed kernel's grid -> no number, and the reason
    live, note = pmc.collect(os.path.join(ROOT, "bench.py"), "c3_4096x3000x44", "physical", "fused", grid_size=12345, timeout_s=30)
    assert live is None and "no dispatch with grid 12345" in note
    # a failing / hanging profiler costs a note, never the run
    monkeypatch.setenv("FAKE_ROCPROF_MODE", "fail")
    live, note = pmc.collect(os.path.join(ROOT, "bench.py"), "c3_4096x3000x44", "physical", "fused", grid_size=3072000, timeout_s=30)
    assert live is None and "exited with 3" in note
    monkeypatch.setenv("FAKE_ROCPROF_MODE", "hang")
    live, note = pmc.collect(os.path.join(ROOT, "bench.py"), "c3_4096x3000x44", "physical", "fused", grid_size=3072000, timeout_s=1.5)
    assert live is None and "did not finish" in note


def test_no_counter_children_under_a_profiler_or_inside_a_child(fake_rocprof, monkeypatch):
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert pmc.collect(os.path.join(ROOT, "bench.py"), "c3_4096x3000x44", "physical", "fused", 3072000)[1] == "this process already runs under a profiler"
    monkeypatch.delenv("ROCPROF_OUTPUT_PATH")
    monkeypatch.setenv("SLGC_BENCH_PMC_CHILD", "1")
    assert pmc.collect(os.path.join(ROOT, "bench.py"), "c3_4096x3000x44", "physical", "fused", 3072000)[1] == "this is a counter child"


def test_scene_registry_and_command_line():
    assert set(bench.SCENES) == {"physical", "noisy-physical", "physical-survey", "s-scene", "s-uniform"}
    for name, cfg in bench.SCENES.items():
        assert cfg["rig"] in ("survey", "covering") and cfg["kind"] in ("physical", "s-scene", "uniform") and cfg["label"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    for flag in ("--scene", "--pmc", "--no-small-images", "--gpus", "--steps", "--warmup", "--extras", "--extras-file"):
        assert flag in out.stdout
    # the driver's defaults: one GPU, the configs[2] workload, the covering-rig capture, counters collected by the run
    import argparse  # noqa: F401
    ns = subprocess.run([sys.executable, "-c", "import sys; sys.argv=['bench.py']; import bench, argparse; "
                         "import re; src=open(bench.__file__).read(); print('default=\"physical\"' in src, 'default=\"auto\"' in src, 'default=\"c3_4096x3000x44\"' in src, 'default=\"lite\"' in src)"],
                        capture_output=True, text=True, cwd=ROOT, timeout=60)
    assert ns.stdout.split() == ["True", "True", "True", "True"], ns.stdout + ns.stderr


def test_printed_line_is_compact_and_carries_the_contract_keys():
    """The driver parses the LAST stdout line: it must stay small whatever the run measured (round 4's 22 KB line was not parsed).  Built here from a
    committed full report of an earlier round (every leg present, long notes) and from a multi-rank report."""
    import json

    from benchlib import line as L
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_steps20.json")))
    assert len(json.dumps(full)) > 20000
    s = L.dump_line(full, "gpurun_out/bench_extras.json")
    assert len(s) < L.LINE_LIMIT and "\n" not in s
    j = json.loads(s)
    assert set(j) == set(L.TOP_KEYS)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert j[k] == full[k], k
    assert set(j["config"]) == set(L.CONFIG_KEYS) and "configs[2]" in j["config"]["workload"] and j["config"]["executed_path"] == "fused"
    assert set(j["roofline"]) == set(L.ROOFLINE_KEYS)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed", "algorithmic_bytes_per_launch"):
        assert j["roofline"][k] == full["roofline"][k], k
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["peak"] == 8000.0 and j["roofline"]["traffic"] > 0
    assert set(j["cpu_baseline"]) == set(L.CPU_KEYS) and len(j["cpu_baseline"]["sample"]) <= 160 and j["cpu_baseline"]["kind"] == "port"
    assert j["cpu_baseline"]["reference_measured"] == 0.046                 # the reference itself rides beside the port's value
    assert j["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and j["cpu_baseline"]["cores"] == 1
    assert j["s_scene_frac"] == full["scenes"]["s-scene"]["frac"] and j["decode_kernel_frac"] == full["decode_kernel_alone"]["roofline"]["frac"]
    assert j["throughput_mode_value"] == full["throughput_mode"]["batched"]["value"]
    assert j["sharded"] is None and j["verify_ok"] is None and j["extras_file"] == "gpurun_out/bench_extras.json"
    assert j["sustained"]["value"] == full["sustained"]["value"] and j["sustained"]["gpu_busy_percent_mean"] == full["sustained"]["gpu"]["gpu_busy_percent_mean"]
    # a multi-rank report: the sharded summary and the verification verdict ride along, long strings are cut
    multi = dict(full, n_gpus=8, sharded={"rccl_nranks": 8, "rccl_rank": 0, "distinct_devices": 8, "rank_devices": [f"0000:{0x15 + 16 * r:02x}:00.0" for r in range(8)], "exchange": "maps", "wire": "hv24", "overlap": True, "with_exchange_value": 1.0, "compute_only_value": 2.0,
                                          "exchange_bytes_per_rank": {"sent": 1, "received": 7}, "compute_only_note": "x" * 5000},
                 verify={"ok": True, "note": "y" * 5000}, error="z" * 5000,
                 sharded_alternatives={"maps_int16": {"value": 5.0, "note": "n" * 900}, "xyz": {"error": "e" * 900}, "maps_hv24_direct": {"value": 7.5}})
    multi["config"] = dict(full["config"], workload="w" * 3000, pipeline="p" * 3000)
    s = L.dump_line(multi, None)
    j = json.loads(s)
    assert len(s) < L.LINE_LIMIT and j["sharded"]["rccl_nranks"] == 8 and j["verify_ok"] is True and set(j["sharded"]) == set(L.SHARDED_KEYS) | {"alternatives"}
    assert j["sharded"]["alternatives"] == {"maps_int16": 5.0, "xyz": "error", "maps_hv24_direct": 7.5}
    # RCCL's own account of the job: rank count from ncclCommCount, one PCI bus id per rank (domain cut), all distinct
    assert j["sharded"]["distinct_devices"] == 8 and j["sharded"]["rank_devices"] == [f"{0x15 + 16 * r:02x}:00.0" for r in range(8)]
    # the S-scene's fraction sits inside the roofline object when the run timed it
    withs = dict(full, roofline=dict(full["roofline"], frac_s_scene=0.7, traffic_over_algorithmic_s_scene=1.147, headline_scene="physical"))
    jr = json.loads(L.dump_line(withs, None))["roofline"]
    assert jr["frac_s_scene"] == 0.7 and jr["traffic_over_algorithmic_s_scene"] == 1.147 and jr["headline_scene"] == "physical"
    assert len(j["config"]["workload"]) <= 120 and len(j["error"]) <= 200
    # a report without any extras (--extras none): the keys are there, empty
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                                 "config", "roofline")}
    j = json.loads(L.dump_line(bare, None))
    assert set(j) == set(L.TOP_KEYS) and j["cpu_baseline"] is None and j["s_scene_frac"] is None and j["roofline"]["frac"] == full["roofline"]["frac"]


def test_launch_statistics_are_robust_to_one_slow_launch():
    """frac comes from the median launch; the mean and the count of launches above 2 x median are beside it (one 10 ms launch among
    twenty of 0.1 ms turned a leg's fraction from 0.72 into 0.12)."""
    from benchlib.common import launch_stats
    st = launch_stats([0.1] * 19 + [10.0])
    assert st["median_launch_ms"] == 0.1 and st["max_launch_ms"] == 10.0 and st["outliers"] == 1
    assert launch_stats([0.1, 0.11, 0.12])["outliers"] == 0 and launch_stats([]) == {}


def test_a_direct_exchange_child_that_dies_costs_an_error_entry_not_the_run(monkeypatch):
    """N > 1: the direct exchange is timed in child processes (tools/benchlib/direct_child.py).  Here there is no GPU: the child cannot even
    create its context -- the parent gets {"error": ...} with the child's exit code and the tail of its stderr, and goes on."""
    import types

    from benchlib import sharded_legs
    args = types.SimpleNamespace(scene="physical", plane_pad=0)
    monkeypatch.setenv("SLGC_BENCH_DIRECT_CHILD_TIMEOUT_S", "120")
    got = sharded_legs.direct_children(args, 0, 2, 0, "cpu_test_key", (64, 48), (64, 48), 26, "hv24", 2, 0, 0x1234, 1, 3)
    assert set(got) == {"error"} and "child of rank 0" in got["error"] and "exit code" in got["error"], got
