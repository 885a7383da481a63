#!/usr/bin/env python3
"""Does the list stage's run-to-run state (NOTES.md: ~200 us or ~260 us per build, steady inside a process) follow the placement of its
buffers?  One process, several allocation sets of the outputs / inputs / workspace, each timed.

  python tools/time_lists_alloc.py [--sets 6] [--iters 30]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from scanner import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3_4096x3000x44")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--sets", type=int, default=6)
args = ap.parse_args()
W, H, pw, ph, N = bench.WORKLOADS[args.workload]
px = W * H
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, pw, ph))
stack = ctx.alloc(N * px)
ctx.synth_scene_dev(stack.ptr, px, N, H, W, seed=1, noise=3, shadow=True)
white_h = np.random.default_rng(9).integers(0, 256, (H, W, 3), dtype=np.uint8)


def timed(maps, white, lists, what):
    def one():
        ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), None, white.ptr, W, H, (pw, ph), lists)
    for _ in range(3):
        one()
    ctx.synchronize()
    ts = []
    for _ in range(3):
        ctx.event_record(0)
        for _ in range(args.iters):
            one()
        ctx.event_record(1)
        ctx.synchronize()
        ts.append(ctx.event_elapsed_ms(0, 1) / args.iters * 1e3)
    print(f"{what:60s} {ts[0]:7.1f} {ts[1]:7.1f} {ts[2]:7.1f} us   cam {lists.cam.ptr:#x} proj {lists.proj.ptr:#x} pts {lists.pts.ptr:#x} col {lists.colors.ptr:#x} maps {maps.ptr:#x}", flush=True)


def new_inputs():
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    white = ctx.alloc(px * 3).upload(white_h)
    ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=_native.TRI_ALGEBRAIC)
    ctx.synchronize()
    xyz.free()
    return maps, white


maps, white = new_inputs()
keep = []
for s in range(args.sets):
    lists = ctx.alloc_cloud_lists(px, colors=True)
    keep.append(lists)
    timed(maps, white, lists, f"output set {s} (earlier sets kept), inputs 0")
for s in range(2):
    timed(maps, white, keep[s], f"output set {s} again")
import copy

if "diag" in os.environ.get("SLGC_LIB", ""):
    names = {0: "all stores", 6: "cam + proj only", 5: "points only", 3: "colours only", 4: "cam + proj + points", 2: "cam + proj + colours", 1: "points + colours", 7: "no stores"}
    for abl, what in names.items():
        os.environ["SLGC_LISTS_ABL"] = str(abl)
        for s in range(len(keep)):
            timed(maps, white, keep[s], f"[{what}] set {s}")
    os.environ["SLGC_LISTS_ABL"] = "0"


def fill_time(ptr, nbytes, iters=20):
    ctx.dev_memset(ptr, 1, nbytes)
    ctx.synchronize()
    ctx.event_record(0)
    for _ in range(iters):
        ctx.dev_memset(ptr, 1, nbytes)
    ctx.event_record(1)
    ctx.synchronize()
    return ctx.event_elapsed_ms(0, 1) / iters * 1e3


for s in range(len(keep)):
    L = keep[s]
    print(f"memset of set {s}: cam {fill_time(L.cam.ptr, px * 8):6.1f} proj {fill_time(L.proj.ptr, px * 8):6.1f} pts {fill_time(L.pts.ptr, px * 24):6.1f} "
          f"colors {fill_time(L.colors.ptr, px * 24):6.1f} us", flush=True)
for field in ("cam", "proj", "pts", "colors"):
    mix = copy.copy(keep[2])
    setattr(mix, field, getattr(keep[0], field))
    timed(maps, white, mix, f"set 2 with {field} of set 0 ({getattr(keep[0], field).ptr:#x})")
class V:
    def __init__(self, ptr):
        self.ptr = ptr


for rep in range(2):
    arena = ctx.alloc(px * 64 + (64 << 20))
    keep.append(arena)
    for skew in (0, 256, 4096, 4096 + 256, 65536 + 4096, (1 << 20) + 4096 + 256, (2 << 20), (6 << 20) + 8192 + 512):
        mix = copy.copy(keep[2])
        off = 0
        for k, (field, size) in enumerate((("cam", px * 8), ("proj", px * 8), ("pts", px * 24), ("colors", px * 24))):
            setattr(mix, field, V(arena.ptr + off + k * skew))
            off += (size + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        timed(maps, white, mix, f"arena {rep}: four outputs carved from one allocation, stream k skewed by k x {skew}")
junk = [ctx.alloc((3 + 2 * i) << 20) for i in range(5)]
lists = ctx.alloc_cloud_lists(px, colors=True)
timed(maps, white, lists, "fresh output set after odd-sized allocations")
