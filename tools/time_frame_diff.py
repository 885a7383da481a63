#!/usr/bin/env python3
"""Times slgc_frame_diff_counts_dev (the count kernel of remove_bad_images) on a stack in HBM with HIP events; counts checked against NumPy.
  python tools/time_frame_diff.py [--workload c3_4096x3000x44] [--iters 40]"""
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
ap.add_argument("--iters", type=int, default=40)
args = ap.parse_args()
W, H, pw, ph, N = bench.WORKLOADS[args.workload]
px = W * H
ctx = _native.Context(0)
stacks = [ctx.alloc(N * px) for _ in range(2)]
for i, s in enumerate(stacks):
    ctx.synth_scene_dev(s.ptr, px, N, H, W, seed=1 + i, noise=12, shadow=True)
cnt = ctx.alloc(8 * N)
for i in range(4):
    ctx.frame_diff_counts_dev(stacks[i % 2].ptr, N, px, 50, cnt.ptr)
ctx.synchronize()
ctx.event_record(0)
for i in range(args.iters):
    ctx.frame_diff_counts_dev(stacks[i % 2].ptr, N, px, 50, cnt.ptr)
ctx.event_record(1)
ctx.synchronize()
us = ctx.event_elapsed_ms(0, 1) / args.iters * 1e3
got = cnt.download((N - 1,), np.uint64)
st = stacks[(args.iters - 1) % 2].download((N, H, W), np.uint8)[:, ::7, ::5].astype(np.int16)
ok = bool((np.abs(np.diff(st, axis=0)) > 50).any()) == bool(got.any())
print(f"{args.workload}: frame differences of {N} frames {us:7.1f} us per call (memset + kernel) = {N * px / us / 1e6:5.2f} TB/s = {N * px / us / 1e6 / 8:.3f} of 8 TB/s | segments "
      f"{os.environ.get('SLGC_FD_SEGS', 'auto')} | sum {int(got.sum())} plausibility {ok}")
