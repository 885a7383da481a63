#!/usr/bin/env python3
"""The kernels of SURVEY.md 8(f)'s "next" rows at the headline size, for `rocprofv3 --kernel-trace --stats` (kernel times do not
include the PCIe legs of the host entry points used here):
  row 1  to_gray             k_bgr_to_gray         4 B / pixel (3 in, 1 out)
  row 3  remove_bad_images   k_frame_diff_count    1 B / pixel / frame (every frame read once)
  row 2  outlier removal     k_knn_* (cloud.hip)   the cloud of the synthetic scan, k = 20

  rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/time_next_rows.py [--frames 44] [--knn-points 2000000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3dscanner-graycode_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from scanner import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3_4096x3000x44")
ap.add_argument("--frames", type=int, default=0, help="frames for the transition filter (default: the workload's N)")
ap.add_argument("--knn-points", type=int, default=0, help="subsample the cloud to this many points (0 = all)")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
W, H, pw, ph, N = bench.WORKLOADS[args.workload]
px = W * H
F = args.frames or N
ctx = _native.Context(0)
ctx.set_calibration(*bench.calibration(W, H, pw, ph))
rng = np.random.default_rng(0)

# row 3: the scan's own frames (device generator), downloaded once and pushed through the host entry point
stack = ctx.alloc(N * px)
ctx.synth_scene_dev(stack.ptr, px, N, H, W, seed=1, noise=3, shadow=True)
frames = stack.download((N, H, W), np.uint8)[:F]
for _ in range(args.reps):
    t = time.perf_counter()
    counts = ctx.frame_diff_counts(frames, 50)
    dt = time.perf_counter() - t
print(f"frame_diff_counts: {F} frames of {W}x{H}: {dt * 1e3:.1f} ms incl. PCIe; counts[:4] = {counts[:4].tolist()}", flush=True)
ref = [(np.abs(frames[j + 1].astype(np.int16) - frames[j].astype(np.int16)) > 50).sum() for j in range(3)]
assert counts[:3].tolist() == [int(x) for x in ref], (counts[:3], ref)

# row 1: 4 BGR frames in one call
bgr = rng.integers(0, 256, (4, H, W, 3), dtype=np.uint8)
for _ in range(args.reps):
    t = time.perf_counter()
    gray = ctx.to_gray(bgr)
    dt = time.perf_counter() - t
y = (bgr[0, :64].astype(np.int64) @ np.array([3735, 19235, 9798]) + (1 << 14)) >> 15
assert np.array_equal(gray[0, :64], y.astype(np.uint8))
print(f"to_gray: 4 frames of {W}x{H}: {dt * 1e3:.1f} ms incl. PCIe", flush=True)

# row 2: k-NN mean distance on the scan's cloud
xyz = ctx.alloc(px * 12)
ctx.scan_dev(stack.ptr, 1, N * px, px, N, H, W, 0, (pw, ph), xyz.ptr, None, mode=_native.TRI_ALGEBRAIC)
cloud = xyz.download((px, 3), np.float32)
cloud = cloud[np.isfinite(cloud[:, 0])]
n_all = len(cloud)
cloud = cloud[(np.abs(cloud) < 0.5).all(axis=1)]             # filter_3d_pts(threshold=0.5) runs first in the reference (src/4-triangulate.py:71)
print(f"cloud: {n_all} points, {len(cloud)} inside the 0.5 box; extents {np.ptp(cloud, axis=0)}", flush=True)
if args.knn_points and args.knn_points < len(cloud):
    cloud = cloud[rng.choice(len(cloud), args.knn_points, replace=False)]
for _ in range(max(1, args.reps - 1)):
    t = time.perf_counter()
    avg = ctx.knn_mean_distance(cloud, 20)
    dt = time.perf_counter() - t
print(f"knn_mean_distance: {len(cloud)} points, k = 20: {dt * 1e3:.1f} ms incl. PCIe, mean {avg.mean():.4e}", flush=True)
