#!/usr/bin/env python3
"""Times the reference-shaped list stage (slgc_cloud_lists_dev: x-major cam_pts / proj_pts, float64 (3,M) points, colours) alone on maps +
XYZ left in HBM by one fused scan; xcd knob interleaved.  Under `rocprofv3 --kernel-trace --stats` the per-kernel split.

  python tools/time_lists.py [--workload c3_4096x3000x44] [--iters 50] [--rounds 4]"""
import argparse
import hashlib
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
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--knobs", default="route=0,1", help="a slgc_tune knob, or route=0,1: 0 = the points come from the fused scan's dense XYZ, "
                                                     "1 = triangulated inside the list build (slgc_cloud_dev's second half); both give the same lists")
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
name, vals = args.knobs.split("=")
vals = [int(v) for v in vals.split(",")]


route = int(os.environ.get("LISTS_ROUTE", "0"))


def one():
    ctx.cloud_lists_dev(maps.at(0), maps.at(px * 2), xyz.ptr if route == 0 else None, white.ptr, W, H, (pw, ph), lists)


res = {v: [] for v in vals}
dig = {}
for r in range(args.rounds):
    for v in vals:
        if name == "route":
            route = v
        else:
            ctx.tune(name, v)
        for _ in range(3):
            one()
        ctx.synchronize()
        ctx.event_record(0)
        for _ in range(args.iters):
            one()
        ctx.event_record(1)
        ctx.synchronize()
        res[v].append(ctx.event_elapsed_ms(0, 1) / args.iters * 1e3)
        if r == 0:
            M = lists.total()
            cam, proj, pts, col = lists.download()
            hsh = hashlib.blake2b(digest_size=8)
            for a in (cam, proj, pts, col):
                hsh.update(np.ascontiguousarray(a).view(np.uint8).data)
            dig[v] = (M, hsh.hexdigest())
M = dig[vals[0]][0]
nodes = ctx.ray_table_info()[0]
for v in vals:
    tri = (v if name == "route" else route) == 1
    # maps twice (count + scatter) 8, white 3, and the dense XYZ 12 -- or, triangulating in-kernel, the camera rays 2 (nodes) / 8 per pixel in; 64 per point out
    nbytes = px * (8 + 3 + ((2 if nodes else 8) if tri else 12)) + M * 64
    t = np.array(res[v])
    print(f"{args.workload} {name}={v}: list stage median {np.median(t):7.1f} us  min {t.min():7.1f}  | {nbytes / np.median(t) / 1e6:6.2f} TB/s "
          f"= {nbytes / np.median(t) / 1e6 / 8:.3f} of 8 TB/s | points {dig[v][0]} digest {dig[v][1]}")
if len({d for d in dig.values()}) != 1:
    print("RESULTS DIFFER")
    sys.exit(1)
