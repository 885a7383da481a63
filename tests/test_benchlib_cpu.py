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
    for flag in ("--scene", "--pmc", "--no-small-images", "--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
    # the driver's defaults: one GPU, the configs[2] workload, the covering-rig capture, counters collected by the run
    import argparse  # noqa: F401
    ns = subprocess.run([sys.executable, "-c", "import sys; sys.argv=['bench.py']; import bench, argparse; "
                         "import re; src=open(bench.__file__).read(); print('default=\"physical\"' in src, 'default=\"auto\"' in src, 'default=\"c3_4096x3000x44\"' in src)"],
                        capture_output=True, text=True, cwd=ROOT, timeout=60)
    assert ns.stdout.split() == ["True", "True", "True"], ns.stdout + ns.stderr
