#!/usr/bin/env python3
"""Time decode-kernel variants on the GPU (HIP events on the launch stream).  Usage:
   python tools/sweep_decode.py [--workload c3] [--variants 16256,8256,...] [--iters 30]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
from scanner import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,4128,104064,104128,104256,108064,108128,108256,116128,116256,1104128,1108128,1108256")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--size", default="4096x3000")
ap.add_argument("--frames", type=int, default=44)
ap.add_argument("--buffers", type=int, default=2)
args = ap.parse_args()
W, H = (int(x) for x in args.size.split("x"))
N = args.frames
ctx = _native.Context(0)
plane = W * H
stacks = []
for b in range(args.buffers):
    s = ctx.alloc(N * plane)
    ctx.synth_scene_dev(s.ptr, plane, N, H, W, seed=1 + b)
    stacks.append(s)
out = ctx.alloc(plane * 4)
ref = None
for v in [int(x) for x in args.variants.split(",")]:
    try:
        for i in range(3):
            ctx.decode_dev(stacks[i % len(stacks)].ptr, 1, N * plane, plane, N, H, W, out.at(0), out.at(plane * 2), variant=v)
        ctx.synchronize()
        ctx.prof_begin(args.iters + 1)
        for i in range(args.iters):
            ctx.decode_dev(stacks[i % len(stacks)].ptr, 1, N * plane, plane, N, H, W, out.at(0), out.at(plane * 2), variant=v)
        ms, n = ctx.prof_end()
        ctx.decode_dev(stacks[0].ptr, 1, N * plane, plane, N, H, W, out.at(0), out.at(plane * 2), variant=v)
        ctx.synchronize()
        got = out.download((2, H, W), np.int16)
        if ref is None:
            ref = got
        same = bool(np.array_equal(ref, got))
        us = ms / n * 1e3
        gbs = (N + 4) * plane / (us * 1e-6) / 1e9
        print(f"variant {v:6d}: {us:8.1f} us  {gbs:7.0f} GB/s  {gbs / 80:5.1f}% of 8 TB/s  same={same}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"variant {v}: {type(e).__name__}: {e}", flush=True)
ctx.close()
