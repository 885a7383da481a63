#!/usr/bin/env python3
"""The scan straight from BGR frames (slgc_scan_bgr_dev) against slgc_to_gray_dev + slgc_scan_dev, interleaved in one process; kernel time of the fused-BGR
kernel from HIP events bound to its dispatch.   usage: time_ingest.py [--workload c3_4096x3000x44] [--iters 30] [--rounds 3]"""
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
ap.add_argument("--scene", default="physical", choices=sorted(bench.SCENES))
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
W, H, pw, ph, N = bench.WORKLOADS[a.workload]
px = W * H
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, pw, ph, rig=bench.SCENES[a.scene]["rig"]))
gray = ctx.alloc(N * px)
caps = []
for b in range(2):
    bench.synth_into(ctx, a.scene, gray.ptr, px, N, H, W, (pw, ph), 31 + b)
    c = ctx.alloc(3 * N * px)
    ctx.synth_bgr_dev(gray.ptr, px, N, H, W, c.ptr, 3 * px)
    caps.append(c)
maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
lib = _native.lib()


def fused(i):
    ctx.scan_bgr_dev(caps[i % 2].ptr, 1, 3 * N * px, 3 * px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2))


def separate(i):
    ctx._ck(lib.slgc_to_gray_dev(ctx._h, caps[i % 2].ptr, N * px, 15, gray.ptr))
    ctx.scan_dev(gray.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, maps.at(0), maps.at(px * 2))


res = {"fused": [], "separate": []}
ks = []
for r in range(a.rounds):
    for name, fn in (("fused", fused), ("separate", separate)):
        for i in range(3):
            fn(i)
        ctx.synchronize()
        if name == "fused":
            ctx.prof_begin(a.iters + 1, 1)
        ctx.event_record(0)
        for i in range(a.iters):
            fn(i)
        ctx.event_record(1)
        if name == "fused":
            ctx.prof_end()
            ks.extend(ctx.prof_samples().tolist())
        res[name].append(ctx.event_elapsed_ms(0, 1) / a.iters)
path = ctx.last_scan_path()["path"]
fused(0)
ctx.synchronize()
kmed = float(np.median(ks)) * 1e3
per = 3 * N + 12
print(f"{a.workload} {a.scene}: scan from BGR frames, fused kernel median {kmed:.1f} us = {per * px / (kmed * 1e-6) / 8e12:.3f} of 8 TB/s on 3N + 12 = {per} B/px; "
      f"per scan fused {np.median(res['fused']) * 1e3:.1f} us vs to_gray + scan {np.median(res['separate']) * 1e3:.1f} us "
      f"({np.median(res['separate']) / np.median(res['fused']):.2f} x)   [{ctx.last_scan_path()['path']}, separate chain last ran '{path}']")
ctx.close()
