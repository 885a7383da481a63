#!/usr/bin/env python3
"""Fused scan with R captures of the same scene max-merged (src/3-capture_decode.py:78-79,95-96; the reference uses R = 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd")); sys.path.insert(0, ROOT)
from scanner import _native
import bench
W, H, PW, PH, N = 4096, 3000, 1920, 1200, 44
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, PW, PH))
px = W * H
for R in (1, 2, 3):
    bufs = []
    for b in range(2):
        s = ctx.alloc(R * N * px)
        for r in range(R):
            ctx.synth_scene_dev(s.at(r * N * px), px, N, H, W, seed=1 + b, noise=3 + 2 * r)
        bufs.append(s)
    maps, xyz = ctx.alloc(px * 4), ctx.alloc(px * 12)
    for mode, name in ((1, "fused"), (1 | 4, "split")):
        for i in range(5):
            ctx.scan_dev(bufs[i % 2].ptr, R, N * px, px, N, H, W, 0, (PW, PH), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
        ctx.synchronize()
        K = 100
        ctx.prof_begin(K + 1, 1)
        for i in range(K):
            ctx.scan_dev(bufs[i % 2].ptr, R, N * px, px, N, H, W, 0, (PW, PH), xyz.ptr, None, maps.at(0), maps.at(px * 2), mode=mode)
        ms, n = ctx.prof_end()
        us = ms / n * 1e3
        byt = (R * N + (16 if name == "fused" else 4)) * px
        print(f"runs {R} {name:6s}: decode kernel {us:7.1f} us per scan  {byt / us / 1e3:7.0f} GB/s on algorithmic bytes ({byt / us / 8e6 * 100:4.1f} % of 8 TB/s)", flush=True)
    for b in bufs + [maps, xyz]:
        b.free()
ctx.close()
