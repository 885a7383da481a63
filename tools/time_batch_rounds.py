#!/usr/bin/env python3
"""How much of a small image's kernel time is the exposed head + tail of ONE round of resident waves?  slgc_scan_batch_dev with n = 1, 2, 3, 4, 8
independent scans per launch: per-scan kernel time against n (rotated sets of stacks > Infinity Cache).   usage: time_batch_rounds.py [workload ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from scanner import _native  # noqa: E402

for wl in (sys.argv[1:] or ["c2_1920x1080x44", "c1_1280x720x42"]):
    W, H, pw, ph, N = bench.WORKLOADS[wl]
    px = W * H
    scene = "physical"
    ctx = _native.Context(0)
    ctx.set_calibration(*bench.calibration(W, H, pw, ph, rig=bench.SCENES[scene]["rig"]))
    for n in (1, 2, 3, 4, 8):
        sets = []
        for b in range(max(2, -(-300_000_000 // (n * N * px)))):
            st = ctx.alloc(n * N * px)
            for s in range(n):
                bench.synth_into(ctx, scene, st.at(s * N * px) if s else st.ptr, px, N, H, W, (pw, ph), 1 + 7 * b + s)
            sets.append(st)
        mh, mv, xyz = ctx.alloc(n * px * 2), ctx.alloc(n * px * 2), ctx.alloc(n * px * 12)

        def go(i):
            ctx.scan_batch_dev(sets[i % len(sets)].ptr, n, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, mh.ptr, mv.ptr)

        for i in range(5):
            go(i)
        ctx.synchronize()
        K = 60
        ctx.prof_begin(K + 8, 1)
        for i in range(K):
            go(i)
        ms, k = ctx.prof_end()
        s = np.sort(ctx.prof_samples()) * 1e3
        med = float(np.median(s))
        print(f"{wl} batch of {n}: kernel median {med:7.2f} us = {med / n:6.2f} us per scan, frac {(N + 12) * px * n / (med * 1e-6) / 8e12:.3f}  ({ctx.last_scan_path()['path']})", flush=True)
        for b in sets + [mh, mv, xyz]:
            b.free()
    ctx.close()
