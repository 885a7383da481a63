#!/usr/bin/env python3
"""Store ablations of the list build's scatter (diagnostic build, wrong results on purpose): which streams cost what.
  SLGC_LIB=3dscanner-graycode_amd/lib/libslgc_diag.so python tools/time_lists_abl.py [--lines 0|1]"""
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
args = ap.parse_args()
W, H, pw, ph, N = bench.WORKLOADS[args.workload]
px = W * H
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, pw, ph))
stack = ctx.alloc(N * px)
ctx.synth_scene_dev(stack.ptr, px, N, H, W, seed=1, noise=3, shadow=True)
maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
white = ctx.alloc(px * 3).upload(np.random.default_rng(9).integers(0, 256, (H, W, 3), dtype=np.uint8))
ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=_native.TRI_ALGEBRAIC)
lists = ctx.alloc_cloud_lists(px, colors=True)
names = {0: "all stores", 6: "cam + proj only", 5: "points only", 3: "colours only", 1: "points + colours", 7: "no stores"}
for lines in (0, 1):
    ctx.tune("lists_lines", lines)
    for abl, what in names.items():
        os.environ["SLGC_LISTS_ABL"] = str(abl)
        ts = []
        for rep in range(3):
            for _ in range(3):
                ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), None, white.ptr, W, H, (pw, ph), lists)
            ctx.synchronize()
            ctx.event_record(0)
            for _ in range(args.iters):
                ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), None, white.ptr, W, H, (pw, ph), lists)
            ctx.event_record(1)
            ctx.synchronize()
            ts.append(ctx.event_elapsed_ms(0, 1) / args.iters * 1e3)
        print(f"lists_lines={lines} [{what:18s}] list stage {min(ts):7.1f} us", flush=True)
